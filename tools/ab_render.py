"""A/B runs inside ONE process: every scene is generated and uploaded once, then rendered under each environment setting in turn
(switches mi_render reads per call: MIW_BVH8, MIW_SHADE_VOTE, MIW_ENV_TOP, MIW_TAIL_PRIO ...), interleaved `--reps` times so that
clock drift hits every variant alike.

    python tools/ab_render.py --scenes matball:256,interior:64 --set "" --set MIW_BVH8=0 [--reps 3] [--debug]

Prints one line per (scene, setting): Msamples/s of each repetition (wall clock around mi_render, film on the device), the path
kernel's HIP-event ms of the last one, tree width, build times. A build-time switch (MIW_BVH4_FAN, MIWAVE_LIB_DIR ...) needs its own
process: run the tool again under that environment.
"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", default="matball:256,interior:64")
    ap.add_argument("--set", action="append", default=[], help="comma-separated NAME=VALUE pairs of one variant ('' = defaults)")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--quality", type=int, default=0)
    args = ap.parse_args()
    import torch
    from mitsuba2_amd import api, scenes
    W, H = 1920, 1080
    film = torch.zeros(H * W * 5, dtype=torch.float32, device="cuda")
    variants = args.set or [""]
    for spec in args.scenes.split(","):
        name, spp = spec.split(":"); spp = int(spp)
        if name == "interior":
            scene, sensor = scenes.interior_scene(W, H, spp, device=-1)
        elif name == "cornell":
            scene, sensor = scenes.cornell_box(W, H, spp, device=-1)
        else:
            scene, sensor = scenes.cornell_box(W, H, spp, diffuse_only=False, device=-1)
        dev = api.Device(0)
        try:
            dev.upload(scene.desc(), bvh_quality=args.quality)
            b = dev.counters()
            print("%s: %d triangles, build %.1f ms (bvh4 %.2f, bvh8 %.2f ms; %d / %d nodes, depth8 %d)" %
                  (name, b.bvh_tris, b.ms_bvh_build, b.ms_bvh4, b.ms_bvh8, b.bvh_nodes, b.bvh8_nodes, b.bvh8_depth), flush=True)
            dev.check(dev.L.mi_set_stream(dev.ctx, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            job = api.PathIntegrator().render_job(sensor)
            cfg = job.cfg
            cfg.film_on_device = 1; cfg.film_f64 = 0; cfg.film_mode = 0; cfg.profile = 1; cfg.plan = 0; cfg.samples_per_launch = spp
            res = {v: [] for v in variants}; last = {}
            for rep in range(args.reps + 1):                       # (repetition 0 warms up)
                for v in variants:
                    kv = dict(p.split("=", 1) for p in v.split(",") if p)
                    known = dev.options()
                    old = {k: (dev.get_option(k) if k in known else os.environ.get(k)) for k in kv}
                    for k, val in kv.items():                       # the library's switches live in the context (read from the environment
                        if k in known:                              # once, by mi_create): mi_set_option; anything else: the environment
                            dev.set_option(k, val)
                        else:
                            os.environ[k] = val
                    try:
                        print("[ab] %s | %s | rep %d" % (name, v or "(defaults)", rep), file=sys.stderr, flush=True)
                        torch.cuda.synchronize(); t0 = time.perf_counter()
                        dev.check(dev.L.mi_render(dev.ctx, C.byref(cfg), C.c_void_p(film.data_ptr())))
                        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
                    finally:
                        for k, o in old.items():
                            if k in known:
                                dev.set_option(k, o)
                            elif o is None:
                                os.environ.pop(k, None)
                            else:
                                os.environ[k] = o
                    if rep:
                        res[v].append(W * H * spp / ms / 1e3)
                    last[v] = dev.counters()
            for v in variants:
                c = last[v]
                print("  %-10s %-34s %s Msamples/s | path kernel %.1f ms, film %.1f ms, width %d, S/sample %.3f" %
                      (name, v or "(defaults)", " ".join("%7.1f" % x for x in res[v]), c.ms_path, c.ms_resolve, c.tree_width, c.segments / max(c.samples, 1)), flush=True)
        finally:
            dev.close()


if __name__ == "__main__":
    main()
