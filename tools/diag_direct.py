import sys, numpy as np
sys.path.insert(0, "/root/repo")
from mitsuba2_amd import api, scenes
scene, sensor = scenes.cornell_box(96, 64, 8, diffuse_only=False, device=-1, ball_level=2)
dev = api.Device(0); dev.upload(scene.desc())
for kw in (dict(), dict(emitter_samples=1, bsdf_samples=0), dict(emitter_samples=0, bsdf_samples=1), dict(emitter_samples=0, bsdf_samples=1, hide_emitters=True)):
    job = api.DirectIntegrator(**kw).render_job(sensor)
    a, st = dev.render(job); ca = dev.counters()
    dev.set_option("MIW_PHASED", "0"); b, st = dev.render(job); cb = dev.counters(); dev.set_option("MIW_PHASED", None)
    d = np.abs(a - b).max(axis=2)
    ys, xs = np.nonzero(d)
    print(kw, "kernels", ca.path_kernel, cb.path_kernel, "differing texels", len(ys), "of", d.size, "segments", ca.segments, cb.segments, "shadow", ca.shadow_rays, cb.shadow_rays)
    if len(ys):
        print("  x range", xs.min(), xs.max(), "y range", ys.min(), ys.max(), "max abs", d.max(), "sum a", a[..., :3].sum(), "sum b", b[..., :3].sum())
        for y, x in list(zip(ys, xs))[:5]:
            print("   ", x, y, a[y, x], b[y, x])
