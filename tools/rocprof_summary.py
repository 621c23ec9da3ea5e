"""Summarise rocprofv3 output directories into the small text files kept under profiles/.

  python tools/rocprof_summary.py stats  <dir-with-*_kernel_trace.csv or *.db>  > profiles/rNN_kernel_stats.txt
  python tools/rocprof_summary.py pmc    <dir> [<dir> ...]                      > profiles/rNN_pmc.txt

`stats` reproduces what `rocprofv3 --kernel-trace --stats` prints per kernel (calls, total, average,
min, max, share); `pmc` sums every counter per kernel over its dispatches and applies the gfx950
corrections of MI355X_MICROARCH.md §HBM (FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE under-reports a
wide coalesced stream by 2x).
"""
import collections
import csv
import glob
import os
import sqlite3
import sys


def _short(name):
    return name.split("(")[0].replace("void ", "")[:48]


def kernel_rows(d):
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if files:
        for f in files:
            for r in csv.DictReader(open(f)):
                yield _short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        return
    for f in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        db = sqlite3.connect(f)
        for name, s, e in db.execute("select name, start, end from kernels"):
            yield _short(name), (e - s) / 1e3


def stats(d):
    agg = collections.defaultdict(list)
    for k, us in kernel_rows(d):
        agg[k].append(us)
    total = sum(sum(v) for v in agg.values())
    print("%-50s %8s %14s %12s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "share"))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("%-50s %8d %14.1f %12.2f %12.2f %12.2f %6.1f%%" % (k, len(v), sum(v), sum(v) / len(v), min(v), max(v),
                                                                 100.0 * sum(v) / total))


def pmc(dirs):
    for d in dirs:
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        n = collections.defaultdict(int)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            seen = set()
            for r in csv.DictReader(open(f)):
                k = _short(r["Kernel_Name"])
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                key = (k, r["Dispatch_Id"])
                if key not in seen:
                    seen.add(key); n[k] += 1
        dur = collections.defaultdict(float)
        for k, us in kernel_rows(d):
            dur[k] += us
        print("== %s" % d)
        for k in sorted(agg, key=lambda k: -dur[k]):
            c = agg[k]
            line = "%-40s dispatches=%d total_ms=%.2f" % (k, n[k], dur[k] / 1e3)
            for name, v in sorted(c.items()):
                line += "  %s=%.5g" % (name, v)
            if "FETCH_SIZE" in c:
                line += "  | hbm_read_GB(corrected x2)=%.3f" % (c["FETCH_SIZE"] * 1024 * 2 / 1e9)
            if "WRITE_SIZE" in c:
                line += "  | hbm_write_GB=%.3f" % (c["WRITE_SIZE"] * 1024 / 1e9)
            print(line)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2:])
