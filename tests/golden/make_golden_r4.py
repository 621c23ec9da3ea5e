"""Regenerates tests/golden/round4.json (run from the repo root: python tests/golden/make_golden_r4.py [c5]).

What round 3 left open (VERDICT r03 "next round" item 1): BASELINE configs[4] — the scalar_spectral Cornell box with a bk7
dielectric block — compared with the oracle AS CONFIGURED: the FULL 1920x1080 frame at 512 spp (1.06e9 samples, 4 wavelengths
each; round 3 compared a 64x64 window). The spectral oracle (oracle/miw_oracle.cpp built with -DMIW_SPECTRAL=1: integrator.cpp:181-288
/ path.cpp:100-211 with spectrum.h:148-314's wavelength sampling and CIE matching) is run HERE once, brute-force scene queries
(34 triangles), and what it produced is committed as digests: sha256 of the float32 film (whole film + per band of 40 rows) and
the sample / segment / shadow-ray counts. tests/test_gpu_configured.py compares the device's film with them. 
  c3    (second session of round 4) BASELINE configs[2] in full as well: the material balls (40 972 triangles, GGX conductor +
        bk7 dielectric), the FULL 1920x1080 frame at the configured 1024 spp (2.12e9 samples; round 3 compared a 128x128 window).
        Oracle scene queries through its own spatial index (orc_set_accel(1); == its brute force: tests/test_oracle_accel.py).
        Hours of host time, once:  python tests/golden/make_golden_r4.py c3 [threads]
Test infrastructure only."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
OUT = os.path.join(ROOT, "tests", "golden", "round4.json")
W, H = 1920, 1080


def main():
    what = set(a for a in sys.argv[1:] if not a.isdigit()) or {"c5"}
    from make_golden_r3 import film_record
    from mitsuba2_amd import api, scenes, build
    build.build_all(oracle=True)
    res = json.load(open(OUT)) if os.path.exists(OUT) else {}
    threads = next((int(a) for a in sys.argv[1:] if a.isdigit()), os.cpu_count() or 8)
    if "c5" in what:
        import oracle_py
        t0 = time.time()
        api.set_variant("scalar_spectral"); api.set_srgb_model(api.default_srgb_coeff())
        try:
            api.host_lib()
            orc = oracle_py.load("scalar_spectral")
            scene, sensor = scenes.cornell_box(W, H, 512, diffuse_only=True, glass_block=True, device=-1)
            job = api.PathIntegrator().render_job(sensor)
            orc.set_accel(0)
            film, _, st = orc.render(scene.desc(), job, threads=threads, want_f64=False)
        finally:
            api.set_variant("scalar_rgb")
        res["c5_full_1920x1080_512spp_spectral"] = film_record(film, st)
        json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)
        print("c5: %.0f s, %d samples, %d segments" % (time.time() - t0, st.samples, st.segments), flush=True)
    if "c3" in what:
        import oracle_py
        t0 = time.time()
        api.host_lib()
        orc = oracle_py.load()
        scene, sensor = scenes.cornell_box(W, H, 1024, diffuse_only=False, device=-1)
        job = api.PathIntegrator().render_job(sensor)
        orc.set_accel(1)
        film, _, st = orc.render(scene.desc(), job, threads=threads, want_f64=False)
        orc.set_accel(0)
        res = json.load(open(OUT)) if os.path.exists(OUT) else res
        res["c3_full_1920x1080_1024spp"] = film_record(film, st)
        json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)
        print("c3: %.0f s, %d samples, %d segments" % (time.time() - t0, st.samples, st.segments), flush=True)
    if "c4blocks" in what:
        # 22 more blocks of the FULL 1920x1080 @ 2048 spp job of config 4 (round 3 pinned the two centre-most): every 97th spiral id.
        # A block's interior (texels >= border = 2 from its edge; clipped blocks: from the image's edge as well) receives that
        # block's samples only, in the full frame as in this shard.
        import numpy as np
        import oracle_py
        from make_golden_r3 import digest, full_job_blocks
        t0 = time.time()
        api.host_lib()
        orc = oracle_py.load()
        scene, sensor = scenes.interior_scene(W, H, 2048, device=-1)
        full = api.PathIntegrator().render_job(sensor)
        ids = list(range(2, int(full.cfg.block_count), 97))
        orc.set_accel(1)
        film, _, st = orc.render(scene.desc(), full, threads=threads, want_f64=False, only_blocks=np.asarray(ids, np.uint32))
        orc.set_accel(0)
        film = np.asarray(film); inner = {}
        for (b, x0, y0), sid in zip(full_job_blocks(full.cfg, ids), ids):
            w = min(28, W - 2 - (x0 + 2)); h = min(28, H - 2 - (y0 + 2))     # clipped edge blocks: stay 2 texels inside the image too
            if w <= 0 or h <= 0:
                continue
            tile = film[y0 + 2:y0 + 2 + h, x0 + 2:x0 + 2 + w]
            inner[str(sid)] = dict(block=b, origin=[x0, y0], size=[w, h], sha256=digest(tile), mean_y=float(tile[..., 1].astype(np.float64).mean()))
        res = json.load(open(OUT)) if os.path.exists(OUT) else res
        res["c4_full_job_more_blocks_2048spp"] = dict(samples=int(st.samples), segments=int(st.segments), interiors=inner, oracle_seconds=round(float(st.seconds), 1))
        json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)
        print("c4 blocks: %.0f s, %d blocks, %d samples" % (time.time() - t0, len(inner), st.samples), flush=True)


if __name__ == "__main__":
    main()
