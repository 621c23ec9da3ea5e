"""Device counterpart of tools/fuzz_cpu.py: the recipes of the CPU fuzzer rendered on the GPU (both tree builders, both plans)
against the scalar restatement. Run by hand on a GPU box:

    python tools/fuzz_gpu.py [--seeds 20] [--first 0]

Seen green on hardware in round 3; sixty of its recipes (seeds 2000 - 2059) are part of the `-m gpu` tier with the oracle's films
committed as digests (tests/test_gpu_configured.py::test_fuzz_recipes_on_the_device). This script remains for longer hand runs
with the oracle live. Test infrastructure only.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=20)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--variant", default="scalar_rgb", choices=["scalar_rgb", "scalar_spectral"],
                    help="scalar_spectral: libmiwave_spectral.so against the spectral restatement (resident plan only: that build has no HBM-queue plan)")
    a = ap.parse_args()
    from mitsuba2_amd import api, scenes, build
    build.build_all(oracle=True)
    spectral = a.variant == "scalar_spectral"
    if spectral:
        api.set_variant("scalar_spectral"); api.set_srgb_model(api.default_srgb_coeff())
    api.host_lib()
    import fuzz_cpu
    import oracle_py
    orc = oracle_py.load(a.variant)
    bad = 0
    dev = api.Device(0)
    for seed in range(a.first, a.first + a.seeds):
        try:
            scene, sensor, ikw, recipe, keep = fuzz_cpu.make_case(api, scenes, seed)
        except Exception as e:
            print("seed %d: recipe rejected: %s" % (seed, str(e)[:120])); continue
        ikw = dict(ikw); ikw.pop("samples_per_pass", None)
        integ = api.DirectIntegrator if ikw.pop("integrator", "path") == "direct" else api.PathIntegrator
        job = integ(**ikw).render_job(sensor)
        o32, _, ost = orc.render(scene.desc(), job, threads=os.cpu_count() or 8, want_f64=False)
        for quality in (0, 0x40):
            dev.upload(scene.desc(), bvh_quality=quality)
            for plan in ((2,) if job.cfg.integrator == 1 or spectral else (2, 1)):     # the direct integrator runs on the resident plan
                g, st = dev.render(job, plan=plan)
                c = dev.counters()
                ok = st == 0 and c.samples == ost.samples and c.segments == ost.segments and np.array_equal(g, o32)
                if not ok:
                    bad += 1
                    print("seed %d MISMATCH (bvh quality %d, plan %d, status %d, segments %d vs %d)\n    %s" %
                          (seed, quality, plan, st, c.segments, ost.segments, "\n    ".join(recipe)))
    dev.close()
    print("%s: %d seeds, %d mismatches" % (a.variant, a.seeds, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
