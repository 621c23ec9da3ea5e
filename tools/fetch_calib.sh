#!/bin/bash
# FETCH_SIZE calibration for the phase machine's access pattern. On the GPU box:  bash tools/fetch_calib.sh > gpurun_out/<tag>_fetch_calib.txt
# Every case runs under rocprofv3 twice (counters in their own passes, --kernel-trace only): FETCH_SIZE, then the L2's hit / miss / memory-side request counters.
cd "$(dirname "$0")/.." && root=$(pwd); export TMPDIR=/tmp
bin=$root/tools/ubench/fetch_calib
[ -x $bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $root/tools/ubench/fetch_calib.hip -o $bin || exit 1
sum_counter() { python3 - "$1" "$2" <<'PY'
import csv, glob, sys
tot = 0.0
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == sys.argv[2] and ("k_scatter" in r["Kernel_Name"] or "k_stream" in r["Kernel_Name"]):
            tot += float(r["Counter_Value"])
print("%.0f" % tot)
PY
}
echo "# mode table_MiB | requested bytes | 64-B sectors touched | 128-B lines touched | FETCH_SIZE x 1024 (as reported) | reported / requested | reported / sectors | reported / lines | TCC hit rate | TCC_EA0_RDREQ x 64 | kernel ms (unprofiled run)"
for case in "stream 1024" "stream 4096" "scatter80 8" "scatter80 128" "scatter80 1024" "scatter80 8192" "scatter48 128" "scatter48 8192"; do
  set -- $case
  line=$($bin $1 $2 256)
  d=$(mktemp -d /tmp/fc_XXXX)
  (cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $d/p0 -- $bin $1 $2 256 > /dev/null 2>&1)
  (cd /tmp && rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $d/p1 -- $bin $1 $2 256 > /dev/null 2>&1)
  (cd /tmp && rocprofv3 --pmc TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $d/p2 -- $bin $1 $2 256 > /dev/null 2>&1)
  fetch=$(sum_counter $d/p0 FETCH_SIZE); hit=$(sum_counter $d/p1 TCC_HIT_sum); miss=$(sum_counter $d/p1 TCC_MISS_sum); rd=$(sum_counter $d/p2 TCC_EA0_RDREQ_sum)
  python3 - "$line" "$fetch" "$hit" "$miss" "$rd" <<'PY'
import sys
w = sys.argv[1].split(); f = float(sys.argv[2]) * 1024.0; hit = float(sys.argv[3]); miss = float(sys.argv[4]); rd = float(sys.argv[5]) * 64.0
req, sec, lin, ms = float(w[4]), float(w[6]), float(w[8]), float(w[10])
print("%-10s %5s | %.4g | %.4g | %.4g | %.4g | %.3f | %.3f | %.3f | %.3f | %.4g | %.3f" % (w[0], w[2], req, sec, lin, f, f / req, f / sec, f / lin, hit / max(hit + miss, 1.0), rd, ms))
PY
  rm -rf $d
done
