"""Turn one tools/profile_round.sh session (gpurun_out/<tag>_*) into the text summary kept under profiles/.

    python tools/make_profile_summary.py r01e profiles/r01_head_c2_kernel_stats_pmc.txt "title line"
"""
import glob
import json
import subprocess
import sys


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def per_frame(j):
    return {k: round(v / max(1, j["steps"]), 2) for k, v in j["roofline"]["kernel_ms"].items()}


def main():
    tag, out_path = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else ""
    g = "gpurun_out/%s" % tag
    out = ["# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline   (tools/profile_round.sh %s)" % tag,
           "# %s (kernel_stats.csv verbatim)" % title,
           open(glob.glob(g + "_trace/*/*kernel_stats.csv")[0]).read().strip(), "",
           "# PMC passes (each counter set in its own run; FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE x2 per MI355X_MICROARCH.md §HBM)",
           subprocess.run([sys.executable, "tools/rocprof_summary.py", "pmc"] + [g + "_pmc%d" % i for i in (1, 2, 3, 4)],
                          capture_output=True, text=True).stdout.strip(), "",
           "# bench lines of the same session (python bench.py ...; ms per frame by kernel from the library's HIP events)"]
    for name in ("c2", "c3", "c5", "c4", "c4_lbvh"):
        j = last_json(g + "_bench_%s.log" % name)
        out.append("%s: %.1f Msamples/s, %.1f ms/frame, kernels ms/frame %s" % (name, j["value"], j["ms_per_step"], per_frame(j)))
        if name == "c2":
            out.append("    roofline " + json.dumps(j["roofline"]))
            out.append("    cpu_baseline " + json.dumps(j["cpu_baseline"]))
    out.append("# shard-size table (bench.py --shard-of N: rank 0's share of an N-GPU job on one GPU), ms/frame")
    for n in (1, 2, 4, 8):
        j = last_json(g + "_shard_%d.log" % n)
        out.append("1/%d: %.1f ms  %s" % (n, j["ms_per_step"], per_frame(j)))
    open(out_path, "w").write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
