"""Diagnostic (GPU box): render the Cornell box at W x H x SPP with the resident plan and with the HBM-queue wavefront
plan and report every texel where the two float32 films differ. This is the tool that found the phantom
Moeller-Trumbore hits described in DESIGN.md section 2 (one sample in 1.3e8).

    python tools/diag_plans.py 1920 1080 64
"""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from mitsuba2_amd import api as native, scenes
W, H, SPP = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
scene, sensor = scenes.cornell_box(W, H, SPP, device=-1)
dev = native.Device(0); dev.upload(scene.desc())
job = native.PathIntegrator().render_job(sensor)
a, _ = dev.render(job, plan=2); c2 = dev.counters()
b, _ = dev.render(job, plan=1); c1 = dev.counters()
print("samples", c2.samples, c1.samples, "segments", c2.segments, c1.segments, "shadow", c2.shadow_rays, c1.shadow_rays)
d = (a != b).any(-1)
print("differing texels", d.sum(), "of", d.size)
if d.any():
    ys, xs = np.nonzero(d)
    print("bbox", xs.min(), xs.max(), ys.min(), ys.max())
    k = np.argmax(np.abs(a - b).max(-1)[d])
    print("max abs diff", np.abs(a - b).max(), "at", xs[k], ys[k], a[ys[k], xs[k]], b[ys[k], xs[k]])
    print("first few", list(zip(xs[:10], ys[:10])))
    a64, _ = dev.render(job, plan=2, f64=True, film_mode=2); b64, _ = dev.render(job, plan=1, f64=True, film_mode=2)
    print("f64 films equal as f32:", np.array_equal(a64.astype(np.float32), b64.astype(np.float32)), "max diff", np.abs(a64 - b64).max())
