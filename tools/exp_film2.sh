cd $GRAFT_REPO_ROOT
export MIW_BENCH_NO_LIVE=1
show='import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],2), d["roofline"]["kernel_ms"])'
for v in "MIW_FL_U=4" "MIW_FL_U=8" "MIW_FL_U=4" "MIW_FL_U=8"; do env $v timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 2>/dev/null | python -c "$show" "$v"; done
for v in "MIW_FL_U=4" "MIW_FILM_LANES=0"; do
env $v timeout 300 python bench.py --no-cpu-baseline --no-extras --scene matball --spp 256 --steps 1 --warmup 1 2>/dev/null | python -c "$show" "matball $v"
env $v timeout 300 python bench.py --no-cpu-baseline --no-extras --scene interior --spp 32 --steps 1 --warmup 1 2>/dev/null | python -c "$show" "interior $v"
done
