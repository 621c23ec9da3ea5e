"""Experiment builds of libmiwave.so with other -D switches, side by side, for A/B runs on the GPU box.

    python tools/build_variants.py name1:-DMIW_WALK=0 name2:-DMIW_WALK=2,-DMIW_LDS_TOP=0 ...

Each variant lands in build_exp/<name>/ (libmiwave.so + a copy of the host library, which finds it through its
$ORIGIN rpath); select one with MIWAVE_LIB_DIR=build_exp/<name>. build_exp/ is git-ignored but travels with gpurun.
Variants are compiled in parallel (one hipcc each)."""
import os
import shutil
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mitsuba2_amd import build  # noqa: E402


def main():
    root = os.path.join(build.ROOT, "build_exp")
    procs = []
    for spec in sys.argv[1:]:
        name, _, defs = spec.partition(":")
        out = os.path.join(root, name)
        os.makedirs(out, exist_ok=True)
        cmd = [build.HIPCC] + build.HIP_FLAGS + [d for d in defs.split(",") if d] + \
              [os.path.join(build.PKG, "csrc", "miwave.hip"), "-o", os.path.join(out, "libmiwave.so")]
        print("[variant %s]" % name, " ".join(cmd), flush=True)
        procs.append((name, out, subprocess.Popen(cmd)))
    bad = 0
    for name, out, p in procs:
        if p.wait() != 0:
            print("variant %s FAILED" % name); bad += 1
            continue
        shutil.copy(os.path.join(build.LIB, "libmiwave_host.so"), out)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
