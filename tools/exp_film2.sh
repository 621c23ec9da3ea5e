cd $GRAFT_REPO_ROOT
export MIW_BENCH_NO_LIVE=1
show='import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],2), d["roofline"]["kernel_ms"])'
for n in 8 4 2; do for v in "MIW_FILM_LANES=1" "MIW_FILM_LANES=0"; do env $v timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 --shard-of $n 2>/dev/null | python -c "$show" "shard 1/$n $v"; done; done
for v in "MIW_FILM_LANES=1" "MIW_FILM_LANES=0"; do env $v timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 --width 1280 --height 720 2>/dev/null | python -c "$show" "720p $v"; done
