# PMC passes over the film replay kernel of one C2 frame (MIW_FILM_QUADS from the environment)
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/film_pmc_${1:-x}
mkdir -p $out
export TMPDIR=/tmp MIW_BENCH_NO_LIVE=1
B="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $B > $out/trace.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $out/pmc1 -- $B > $out/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $out/pmc2 -- $B > $out/pmc2.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $out/pmc3 -- $B > $out/pmc3.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE  --kernel-trace --output-format csv -d $out/pmc4 -- $B > $out/pmc4.log 2>&1
find $out -name "*.db" -delete 2>/dev/null
cd $out
python - <<'PY'
import csv, glob, collections
for d in ("pmc1", "pmc2", "pmc3", "pmc4"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if "film" in r["Kernel_Name"] and "merge" not in r["Kernel_Name"]:
                acc[(r["Kernel_Name"][:40], r["Counter_Name"])] += float(r["Counter_Value"])
        for k, v in sorted(acc.items()):
            print(d, k[0], k[1], "%.6g" % v)
for f in glob.glob("trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:60], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
PY
