"""Regenerates tests/golden/round3.json (run from the repo root: python tests/golden/make_golden_r3.py [c2] [c3] [c4] [fuzz]).

What round 2 left open (VERDICT r02 "next round" item 1): the BASELINE configurations compared with the oracle AS CONFIGURED.
The oracle (oracle/miw_oracle.cpp, the scalar restatement of integrator.cpp:181-288 / path.cpp:100-211) is run HERE, on the host
cores of the build container, once, and what it produced is committed as digests: sha256 of the float32 film bytes (whole film
and per band of rows, so that a mismatch can be localised) + the sample / segment / shadow-ray counts. The GPU tier compares the
device's films with these digests (tests/test_gpu_configured.py); nothing here needs a GPU.

  c2    BASELINE configs[1] in full: Cornell box 1920x1080 @ 512 spp, diffuse (1.06e9 samples; ~5 min on 8 cores)
  c3    configs[2]: material balls, 1080p sensor, a 128x128 window at the full 1024 spp over both balls' silhouettes
        (oracle scene queries through its own BVH, orc_set_accel(1); proven == its brute force in tests/test_oracle_accel.py)
  c4    configs[3]: 911 362-triangle interior, area light + environment map, two 64x32 windows at the configured 2048 spp
        (the frame's left edge, where pixels look past the room straight into the environment map; conductor / dielectric /
        diffuse clutter lit through the open ceiling), and a 2-block shard of the FULL 1080p @ 2048 spp job (the blocks' own
        ids and seeds) for the full-size log test
  fuzz  digests of tools/fuzz_cpu.py recipes 2000..2059 (the device fuzz tier's committed answers)
Test infrastructure only."""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
OUT = os.path.join(ROOT, "tests", "golden", "round3.json")
W, H = 1920, 1080
BAND = 40           # rows per band digest (1080 = 27 bands)


def digest(film):
    return hashlib.sha256(np.ascontiguousarray(film, np.float32).tobytes()).hexdigest()


def film_record(film, st, **extra):
    film = np.asarray(film)
    rec = dict(sha256=digest(film), samples=int(st.samples), segments=int(st.segments), shadow_rays=int(st.shadow_rays),
               mean_y=float(film[..., 1].astype(np.float64).mean()), shape=list(film.shape), oracle_seconds=round(float(st.seconds), 1))
    if film.shape[0] > BAND:
        rec["bands"] = [digest(film[y:y + BAND]) for y in range(0, film.shape[0], BAND)]
    rec.update(extra)
    return rec


def crop_job(api, scenes, spp, x, y, w, h, n_threads):
    sensor = scenes.cornell_sensor(W, H, spp, crop_offset_x=x, crop_offset_y=y, crop_width=w, crop_height=h)
    return api.PathIntegrator().render_job(sensor, n_threads=n_threads)


# the windows / shards the GPU tests render (tests/test_gpu_configured.py imports these)
C3_WINDOW = (928, 936, 128, 128)          # x, y, w, h on the 1920x1080 sensor: both balls' silhouettes with the back wall between them
C4_WINDOWS = {"edge": (0, 520, 64, 32),   # the frame's left edge: columns that look past the room into the environment map + the red wall
              "clutter": (736, 584, 64, 32)}   # conductor / dielectric / diffuse spheres in front of the back wall, lit through the open ceiling
C4_FULL_BLOCKS = (0, 1)                   # spiral ids (centre-most first) of the 2-block shard of the full-frame job


def full_job_blocks(cfg, ids):
    """row-major indices and pixel origins of the spiral blocks `ids` of a full-frame job"""
    n = int(cfg.block_count); bs = int(cfg.block_size); nbx = (int(cfg.crop_w) + bs - 1) // bs
    table = [int(cfg.block_ids[i]) for i in range(n)]
    out = []
    for i in ids:
        b = table.index(i)
        out.append((b, (b % nbx) * bs, (b // nbx) * bs))
    return out


def main():
    what = set(sys.argv[1:]) or {"c2", "c3", "c4", "fuzz"}
    from mitsuba2_amd import api, scenes, build
    build.build_all(oracle=True)
    api.host_lib()
    import oracle_py
    orc = oracle_py.load()
    threads = os.cpu_count() or 8
    res = json.load(open(OUT)) if os.path.exists(OUT) else {}

    def save():
        json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)

    if "c2" in what:
        t0 = time.time()
        scene, sensor = scenes.cornell_box(W, H, 512, device=-1)
        job = api.PathIntegrator().render_job(sensor)
        orc.set_accel(0)
        film, _, st = orc.render(scene.desc(), job, threads=threads, want_f64=False)
        res["c2_full_1920x1080_512spp"] = film_record(film, st)
        save(); print("c2: %.0f s, %d samples, %d segments" % (time.time() - t0, st.samples, st.segments), flush=True)
    if "c3" in what:
        t0 = time.time()
        scene, _ = scenes.cornell_box(W, H, 1024, diffuse_only=False, device=-1)
        x, y, w, h = C3_WINDOW
        job = crop_job(api, scenes, 1024, x, y, w, h, n_threads=256)
        orc.set_accel(1)
        film, _, st = orc.render(scene.desc(), job, threads=threads, want_f64=False)
        res["c3_window_1024spp"] = film_record(film, st, window=list(C3_WINDOW))
        save(); print("c3: %.0f s, %d samples, %d segments" % (time.time() - t0, st.samples, st.segments), flush=True)
    if "c4" in what:
        scene, sensor = scenes.interior_scene(W, H, 2048, device=-1)
        orc.set_accel(1)
        for name, (x, y, w, h) in C4_WINDOWS.items():
            t0 = time.time()
            job = crop_job(api, scenes, 2048, x, y, w, h, n_threads=128)
            film, _, st = orc.render(scene.desc(), job, threads=threads, want_f64=False)
            res["c4_window_%s_2048spp" % name] = film_record(film, st, window=[x, y, w, h])
            save(); print("c4 window %s: %.0f s, %d samples, %d segments" % (name, time.time() - t0, st.samples, st.segments), flush=True)
        # two blocks of the FULL 1920x1080 @ 2048 spp job (their own spiral ids and seeds): what the full-size log test compares
        # the interiors of those blocks with (texels at least `border` = 2 away from the block's edge receive this block's
        # samples only, in the full frame as in the shard)
        t0 = time.time()
        full = api.PathIntegrator().render_job(sensor)
        film, _, st = orc.render(scene.desc(), full, threads=threads, want_f64=False, only_blocks=np.asarray(C4_FULL_BLOCKS, np.uint32))
        film = np.asarray(film); inner = {}
        for (b, x0, y0), sid in zip(full_job_blocks(full.cfg, C4_FULL_BLOCKS), C4_FULL_BLOCKS):
            inner[str(sid)] = dict(block=b, origin=[x0, y0], sha256=digest(film[y0 + 2:y0 + 30, x0 + 2:x0 + 30]),
                                   mean_y=float(film[y0 + 2:y0 + 30, x0 + 2:x0 + 30, 1].astype(np.float64).mean()))
        res["c4_full_job_blocks_2048spp"] = dict(samples=int(st.samples), segments=int(st.segments), interiors=inner, oracle_seconds=round(float(st.seconds), 1))
        save(); print("c4 full-job blocks: %.0f s, %d samples" % (time.time() - t0, st.samples), flush=True)
        orc.set_accel(0)
    if "fuzz" in what:
        import fuzz_cpu
        t0 = time.time()
        orc.set_accel(0)
        fz = {}
        for seed in range(2000, 2060):
            try:
                scene, sensor, ikw, recipe, keep = fuzz_cpu.make_case(api, scenes, seed)
            except Exception as e:      # a recipe the host layer refuses (e.g. nested transmissive twosided): skipped on both sides
                fz[str(seed)] = dict(rejected=str(e)[:80]); continue
            ikw = dict(ikw); ikw.pop("samples_per_pass", None)
            integ = api.DirectIntegrator if ikw.pop("integrator", "path") == "direct" else api.PathIntegrator
            job = integ(**ikw).render_job(sensor)
            film, _, st = orc.render(scene.desc(), job, threads=2, want_f64=False)
            fz[str(seed)] = dict(sha256=digest(film), samples=int(st.samples), segments=int(st.segments))
        res["fuzz"] = fz
        save(); print("fuzz: %.0f s, %d recipes" % (time.time() - t0, len(fz)), flush=True)


if __name__ == "__main__":
    main()
