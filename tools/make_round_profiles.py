"""Turn one tools/profile_round.sh session (gpurun_out/<tag>_*) into the files kept under profiles/:

    python tools/make_round_profiles.py r02            # -> profiles/r02_c2_kernel_stats_pmc.txt, r02_c3_tree_..., r02_c4_tree_..., traffic.json

Every summary starts with the hash of the kernel sources it was measured on (bench.kernel_src_sha16()), the rocprofv3
kernel_stats.csv verbatim, the four PMC passes summed per kernel (tools/rocprof_summary.py), then the bench lines of the same
session. Lines of an existing summary after the marker "# notes" (statistics of debug builds from other sessions) are kept.
profiles/traffic.json gets one entry per profiled workload: HBM bytes per launch (FETCH_SIZE KiB x 1024 x 2 + WRITE_SIZE KiB x
1024, MI355X_MICROARCH.md section HBM), VALU issue-slot use, lane use and memory-wait share of every kernel; bench.py copies the
matching entry into its roofline block while the kernel sources still hash to the same value.
"""
import collections
import csv
import glob
import io
import json
import os
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
import rocprof_summary  # noqa: E402


def drop_stale_passes(g):
    """gpurun MERGES a session's files into gpurun_out/: a tag profiled twice leaves two sessions' CSVs (named by process id) in every pass directory and
    every counter would be summed over both. Keep the newest process's files per directory."""
    import re
    for d in glob.glob(g + "_*/**/", recursive=True):
        by = collections.defaultdict(list)
        for f in os.listdir(d):
            m = re.match(r"(\d+)_.*\.csv$", f)
            if m:
                by[m.group(1)].append(os.path.join(d, f))
        if len(by) > 1:
            newest = max(by, key=lambda k: max(os.path.getmtime(f) for f in by[k]))
            for k, fs in by.items():
                if k != newest:
                    for f in fs:
                        os.remove(f)


def last_json(path):
    try:
        return json.loads(open(path).read().strip().splitlines()[-1])
    except Exception:
        return None


def per_frame(j):
    return {k: round(v / max(1, j["steps"]), 2) for k, v in j["roofline"]["kernel_ms"].items()}


def bench_line(g, name, label=None, extra=""):
    j = last_json("%s_%s.log" % (g, name))
    if j is None:
        return "%s: (no line)" % (label or name)
    import re
    m = re.search(r"@ (\d+) spp", j["config"]["workload"])
    return "%s: %.1f Msamples/s, %.1f ms/frame (%s spp), S = %.2f, kernels ms/frame %s [bvh: %s, %.1f ms]%s" % (
        label or name, j["value"], j["ms_per_step"], m.group(1) if m else "?", j["roofline"].get("segments_per_sample", float("nan")), per_frame(j),
        j["config"]["bvh"]["builder"], j["config"]["bvh"]["build_ms"], extra)


def pmc_text(g, name):
    buf = io.StringIO()
    with redirect_stdout(buf):
        rocprof_summary.pmc(["%s_%s_pmc%d" % (g, name, i) for i in (1, 2, 3, 4)])
    return buf.getvalue().strip()


def counters(d):
    """{short kernel name: {counter: sum}} and {kernel: total ms} of one PMC pass directory."""
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[rocprof_summary._short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    dur = collections.defaultdict(float)
    for k, us in rocprof_summary.kernel_rows(d):
        dur[k] += us / 1e3
    return agg, dur


def traffic_entry(g, name, sha, source):
    c = collections.defaultdict(dict)
    ms = {}
    for i in (1, 2, 3, 4):
        agg, dur = counters("%s_%s_pmc%d" % (g, name, i))
        for k, v in agg.items():
            for cn, val in v.items():
                c[k][cn + ("@%d" % i if cn == "SQ_WAVE_CYCLES" else "")] = val
        if i == 1:
            ms = dur
    out = {"kernel_src_sha16": sha, "source": source}
    for k, v in c.items():
        if not k.startswith("k_"):
            continue
        short = k.split("<")[0]
        if "FETCH_SIZE" not in v or "SQ_INSTS_VALU" not in v:
            continue
        rd, wr = v["FETCH_SIZE"] * 1024 * 2, v["WRITE_SIZE"] * 1024
        simd_cycles = 1024.0 * v["GRBM_GUI_ACTIVE"] / 8.0
        out[short] = {
            "hbm_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr,
            "valu_issue_frac": round(v["SQ_INSTS_VALU"] * 2.0 / simd_cycles, 4),
            "lane_util": round(v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_ACTIVE_INST_VALU"]), 4) if v.get("SQ_ACTIVE_INST_VALU") else None,
            "wait_mem_frac": round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES@2"], 4) if v.get("SQ_WAVE_CYCLES@2") else None,
            "ms": round(ms.get(k, 0.0), 3),
        }
    return out


def kept_notes(path):
    if not os.path.exists(path):
        return []
    lines = open(path).read().splitlines()
    for i, l in enumerate(lines):
        if l.startswith("# notes"):
            return lines[i:]
    return []


def main():
    tag = sys.argv[1]
    os.chdir(ROOT)
    g = os.path.join("gpurun_out", tag)
    drop_stale_passes(g)
    sha = bench.kernel_src_sha16()
    head = "# kernel sources sha256[:16] = %s (bench.kernel_src_sha16(): mitsuba2_amd/csrc/**/*.{h,hip})" % sha
    cmd = "# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline%s   (tools/profile_round.sh %s)"
    stats = lambda name: open(glob.glob("%s_%s_trace/*/*kernel_stats.csv" % (g, name))[0]).read().strip()
    pmc_head = "# PMC passes (each counter set in its own run; FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE x2 per MI355X_MICROARCH.md §HBM)"
    traffic = {"_comment": json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["_comment"]}

    # ---- C2 ----
    path = os.path.join(ROOT, "profiles", "%s_c2_kernel_stats_pmc.txt" % tag)
    notes = kept_notes(path)
    out = [head, cmd % ("", tag), "# C2 = Cornell box 1920x1080 @ 512 spp, diffuse (the default bench) — kernel_stats.csv verbatim", stats("c2"), "",
           pmc_head, pmc_text(g, "c2"), "", "# bench lines of the same session (ms per frame by kernel from the library's HIP events)"]
    j = last_json(g + "_bench_c2.log")
    out.append("c2: %.1f Msamples/s, %.1f ms/frame, kernels ms/frame %s" % (j["value"], j["ms_per_step"], per_frame(j)))
    out.append("    roofline " + json.dumps(j["roofline"]))
    out.append("    cpu_baseline " + json.dumps(j["cpu_baseline"]))
    out.append("# the other lines of the session, named as tools/profile_round.sh names them (what each name stands for is written there)")
    for f in sorted(glob.glob(g + "_bench_c2_*.log") + glob.glob(g + "_bench_direct_c2.log") + glob.glob(g + "_bench_c5*.log")):
        out.append(bench_line(g, os.path.basename(f)[len(tag) + 1:-4]))
    out.append("# tile-shard table (bench.py --shard tiles --shard-of N: rank 0's share of an N-GPU job on one GPU), ms/frame")
    for f in sorted(glob.glob(g + "_shard_*.log")):
        j = last_json(f)
        if j:
            out.append("%s: %.2f ms  %s" % (os.path.basename(f)[len(tag) + 1:-4], j["ms_per_step"], per_frame(j)))
    open(path, "w").write("\n".join(out + notes) + "\n")
    traffic["scalar_rgb/cornell/1920x1080@512/plan2/film1/launch512"] = traffic_entry(g, "c2", sha, "profiles/" + os.path.basename(path))

    # ---- C3 / C4: the tree kernels ----
    for name, title, key, spp in (
            ("c3", "C3 geometry = material balls (GGX conductor + bk7 dielectric, 40 972 triangles), 1920x1080 @ 64 spp (profile)", "matball", 64),
            ("c4", "C4 class = procedural interior (911 362 triangles, area light + 1024x512 envmap), 1920x1080 @ 16 spp (profile)", "interior", 16)):
        path = os.path.join(ROOT, "profiles", "%s_%s_tree_kernel_stats_pmc.txt" % (tag, name))
        notes = kept_notes(path)
        out = [head, cmd % (" --scene %s --spp %d" % (key, spp), tag), "# %s — kernel_stats.csv verbatim" % title, stats(name), "",
               pmc_head, pmc_text(g, name), "",
               "# bench lines of the same session, named as tools/profile_round.sh names them (bvh builder and build ms in brackets)"]
        for f in sorted(glob.glob("%s_bench_%s*.log" % (g, name))):
            out.append(bench_line(g, os.path.basename(f)[len(tag) + 1:-4]))
        if name == "c4" and os.path.exists(g + "_bench_c4.err"):
            out += ["#   " + l.strip() for l in open(g + "_bench_c4.err") if "device builder" in l or "bvh4" in l or "bvh8" in l or "LDS per" in l]
        if name == "c4" and os.path.exists(g + "_bench_c4_sah_r4.err"):
            out += ["#   MIW_SAH_HUGE=0 (one workgroup per candidate, round 4's launch shape): " + l.strip() for l in open(g + "_bench_c4_sah_r4.err") if "device builder" in l]
        if name == "c4" and glob.glob(g + "_c4bvh4_pmc1"):
            out += ["", "# the same workload through the 4-wide walk of the same build (MIW_BVH8=0): PMC passes", pmc_text(g, "c4bvh4")]
        if name == "c3":
            out.append("# triangle-count series (bench.py --scene matball --tess t --spp 128): 52 triangles = packet kernel; from 172 on the phase machine")
            for t in range(5):
                j = last_json(g + "_tess_%d.log" % t)
                if j:
                    out.append("tess %d: %6d triangles  %.1f Msamples/s  %s" % (t, j["config"]["bvh"]["tris"], j["value"], j["roofline"]["kernel"]))
        out.append("# rank 0's tile shards on one GPU (names: tools/profile_round.sh), ms/frame")
        for f in sorted(glob.glob("%s_%s_shard_*.log" % (g, name))):
            j = last_json(f)
            if j:
                out.append("%s: %.2f ms  %s" % (os.path.basename(f)[len(tag) + 1:-4], j["ms_per_step"], per_frame(j)))
        open(path, "w").write("\n".join(out + notes) + "\n")
        traffic["scalar_rgb/%s/1920x1080@%d/plan2/film1/launch%d" % (key, spp, spp)] = traffic_entry(g, name, sha, "profiles/" + os.path.basename(path))

    # ---- plan 1 (round 6): the wavefront plan with its SoA queues in HBM, the structure north_star names — its REAL queue traffic ----
    if glob.glob(g + "_c3plan1_trace"):
        path = os.path.join(ROOT, "profiles", "%s_c3_plan1_kernel_stats_pmc.txt" % tag)
        out = [head, cmd % (" --scene matball --spp 64 --plan 1", tag),
               "# plan 1 on the material balls, 1920x1080 @ 64 spp: k_init_lanes, the persistent k_trace_stream + k_sort_hits, k_shade per depth-loop iteration over", 
               "# SoA queues of 16-byte fields in HBM (DESIGN.md section 4.3) — kernel_stats.csv verbatim", stats("c3plan1"), "", pmc_head, pmc_text(g, "c3plan1"), "",
               "# bench line of the same session (256 spp)", bench_line(g, "bench_c3_plan1")]
        try:
            e = traffic_entry(g, "c3plan1", sha, "profiles/" + os.path.basename(path))
            j = last_json(g + "_c3plan1_trace.log") or {}
            tot = sum(v["hbm_bytes_per_launch"] * 0 + v["read_bytes"] + v["write_bytes"] for k, v in e.items() if isinstance(v, dict))
            out += ["", "# real HBM-side bytes of the whole frame, all kernels (FETCH_SIZE x 2 + WRITE_SIZE, summed over the launches): %.1f GB" % (tot / 1e9)]
            traffic["scalar_rgb/matball/1920x1080@64/plan1/film1/launch64"] = e
        except Exception as ex:
            out.append("# (traffic summary failed: %r)" % (ex,))
        open(path, "w").write("\n".join(out) + "\n")

    # every rank's tile shard of an 8-GPU frame + the floor (tools/shard_table.py, same session)
    if os.path.exists(g + "_shards.txt"):
        open(os.path.join(ROOT, "profiles", "%s_shards.txt" % tag), "w").write(head + "\n" + open(g + "_shards.txt").read())
    if os.path.exists(g + "_bench_c2.log"):
        open(os.path.join(ROOT, "profiles", "%s_bench_default_line.json" % tag), "w").write(open(g + "_bench_c2.log").read().strip().splitlines()[-1] + "\n")
    json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print("kernel_src_sha16", sha)


if __name__ == "__main__":
    main()
