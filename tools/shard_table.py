"""What EVERY rank of an N-GPU frame renders, measured on one GPU: per-rank time, the slowest rank (= how long
the N-GPU frame lasts), the mean, and the floor.

    python tools/shard_table.py [--ranks 8] [--configs c2,c3,c4] [--spp-scale 1.0] [--out profiles/r05_shards.txt]

For each configuration the scene is uploaded once; then the full frame and, one after the other, the tile shard of every rank r of
N (spiral blocks with id % N == r, all spp of their pixels: PathIntegrator::set_shard) are rendered through
mi_render exactly as `bench.py --shard-of N --shard-index r` does (film on the device, HIP-event profile on), twice each; the
second run is quoted. An N-GPU frame lasts as long as its slowest rank (+ the < 1 ms film reduce), so
    predicted speed-up = full-frame ms / max-rank ms.
The floor: a shard of about one pixel per resident lane cannot finish before its dearest pixel has run its spp samples one after the
other. The measuring launch of a placed render reports that pixel (mi_counters::place_max_pixel, place_cost_max over the first
place_measure_spp samples); the table quotes (a) cost_max / cost_mean of every rank and (b) the time of a 1 x 1-pixel crop at the
dearest pixel of the slowest rank rendered ALONE at the full spp (other random numbers than inside the frame — a crop has its own
block grid — but the same pixel footprint): no schedule of that rank can beat it.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--configs", default="c2,c3,c4")
    ap.add_argument("--spp-scale", type=float, default=1.0, help="multiplies every configuration's spp (1.0 = as configured)")
    ap.add_argument("--out", default="")
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    import torch
    from mitsuba2_amd import api, scenes
    W, H = 1920, 1080
    film = torch.zeros(H * W * 5, dtype=torch.float32, device="cuda")
    lines = []; table = {}

    def say(s):
        print(s, flush=True); lines.append(s)

    def render(dev, job, spp):
        cfg = job.cfg
        cfg.film_on_device = 1; cfg.film_f64 = 0; cfg.film_mode = 0; cfg.profile = 1; cfg.plan = 0; cfg.samples_per_launch = int(spp)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        dev.check(dev.L.mi_render(dev.ctx, C.byref(cfg), C.c_void_p(film.data_ptr())))
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3, dev.counters()

    specs = {"c2": ("C2 Cornell box, diffuse", 512, lambda spp: scenes.cornell_box(W, H, spp, device=-1)),
             "c3": ("C3 material balls", 1024, lambda spp: scenes.cornell_box(W, H, spp, diffuse_only=False, device=-1)),
             "c4": ("C4 interior 0.9 M triangles + envmap", 2048, lambda spp: scenes.interior_scene(W, H, spp, device=-1))}
    N = args.ranks
    say("# tile shards of a %d-GPU frame, every rank, on one MI355X (tools/shard_table.py; second of two runs each; ms = wall clock around mi_render)" % N)
    for key in args.configs.split(","):
        name, spp0, make = specs[key]
        spp = max(16, int(round(spp0 * args.spp_scale)))
        scene, sensor = make(spp)
        dev = api.Device(0)
        try:
            dev.upload(scene.desc())
            dev.check(dev.L.mi_set_stream(dev.ctx, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            integ = api.PathIntegrator()
            job = integ.render_job(sensor)
            render(dev, job, spp); full_ms, c = render(dev, job, spp)
            say("\n## %s, 1920x1080 @ %d spp — full frame %.1f ms (path kernel %.1f ms, film %.1f ms, tree width %d)" % (name, spp, full_ms, c.ms_path, c.ms_resolve, c.tree_width))
            say("rank   ms      path-kernel  film    samples        placed  cost max / mean (unit)   dearest pixel")
            rows = []
            for r in range(N):
                integ_r = api.PathIntegrator(); integ_r.set_shard(r, N)
                job_r = integ_r.render_job(sensor)
                render(dev, job_r, spp); ms, c = render(dev, job_r, spp)
                px = (int(c.place_max_pixel) & 0xffff, int(c.place_max_pixel) >> 16)
                rows.append(dict(rank=r, ms=ms, ms_path=c.ms_path, ms_film=c.ms_resolve, samples=int(c.samples), placed=int(c.placed),
                                 cost_max=int(c.place_cost_max), cost_mean=float(c.place_cost_mean), unit=int(c.place_cost_unit), pixel=px,
                                 measure_spp=int(c.place_measure_spp)))
                say("%-6d %-7.1f %-12.1f %-7.1f %-14d %-7d %6d / %8.1f (%s)      (%d, %d)" %
                    (r, ms, c.ms_path, c.ms_resolve, c.samples, c.placed, c.place_cost_max, c.place_cost_mean,
                     "256 clk" if c.place_cost_unit else "iter", px[0], px[1]))
            mx = max(rows, key=lambda q: q["ms"]); mean = sum(q["ms"] for q in rows) / N; mn = min(q["ms"] for q in rows)
            say("max-rank %.1f ms (rank %d), mean %.1f ms, min %.1f ms: max / mean = %.3f; predicted %d-GPU speed-up = %.1f / %.1f = %.2fx "
                "(mean-rank figure: %.2fx)" % (mx["ms"], mx["rank"], mean, mn, mx["ms"] / mean, N, full_ms, mx["ms"], full_ms / mx["ms"], full_ms / mean))
            # the floor: the slowest rank's dearest pixel alone, full spp
            floor_ms = None
            if mx["placed"]:
                x, y = mx["pixel"]
                s1 = scenes.cornell_sensor(W, H, spp, crop_offset_x=int(x), crop_offset_y=int(y), crop_width=1, crop_height=1)
                job1 = api.PathIntegrator().render_job(s1)
                render(dev, job1, spp); ms1, c1 = render(dev, job1, spp)
                floor_ms = c1.ms_path
                say("floor: pixel (%d, %d) alone, %d spp: path kernel %.1f ms (%.0f segments) — the slowest rank's path kernel took %.1f ms = %.2fx that; "
                    "its dearest pixel cost %.2fx the rank's mean pixel in the measuring launch" %
                    (x, y, spp, c1.ms_path, c1.segments, mx["ms_path"], mx["ms_path"] / max(c1.ms_path, 1e-9), mx["cost_max"] / max(mx["cost_mean"], 1e-9)))
            table[key] = dict(name=name, spp=spp, full_ms=full_ms, ranks=rows, max_ms=mx["ms"], mean_ms=mean, floor_pixel_alone_ms=floor_ms,
                              predicted_speedup=full_ms / mx["ms"])
        finally:
            dev.close()
    if args.out:
        open(args.out, "w").write("\n".join(lines) + "\n")
    if args.json:
        json.dump(table, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
