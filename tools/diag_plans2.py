import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
from mitsuba2_amd import api as native, scenes
import oracle_py
orc = oracle_py.load()
W, H, SPP = 1920, 1080, 64
scene, sensor = scenes.cornell_box(W, H, SPP, device=-1)
dev = native.Device(0); dev.upload(scene.desc())
job = native.PathIntegrator().render_job(sensor)
a, _ = dev.render(job, plan=2)
b, _ = dev.render(job, plan=1)
nbx = (W + 31) // 32
bid = int(job.block_ids[18 * nbx + 21])
o32, _, st = orc.render(scene.desc(), job, threads=8, want_f64=False, only_blocks=[bid])
for (x, y) in [(700, 595), (701, 595), (700, 596), (701, 596), (690, 590)]:
    print((x, y), "plan2", a[y, x, :3], "plan1", b[y, x, :3], "oracle", o32[y, x, :3],
          "plan2==oracle", np.array_equal(a[y, x], o32[y, x]), "plan1==oracle", np.array_equal(b[y, x], o32[y, x]))
# same with the leaf filter switched off (plan 2 full sweeps) and with the tree walk forced
from mitsuba2_amd import _capi
for q, name in ((1 | 0x20, "no leaf filter"), (1 | 0x10, "forced tree")):
    dev.upload(scene.desc(), bvh_quality=q)
    c, _ = dev.render(job, plan=2)
    print(name, "plan2 == plan2(filter):", np.array_equal(c, a), " == plan1:", np.array_equal(c, b))
