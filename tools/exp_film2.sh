cd $GRAFT_REPO_ROOT
export MIW_BENCH_NO_LIVE=1
show='import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],2), d["roofline"]["kernel_ms"])'
for v in "MIW_FILM_QUADS=24" "MIW_FILM_QUADS=42" "MIW_FILM_QUADS=44" "MIW_FILM_QUADS=0" "MIW_FILM_COLUMNS=0" "MIW_FILM_GROUP=2" "MIW_FILM_QUADS=24" "MIW_FILM_QUADS=42"; do env $v timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --shard-of 8 2>/dev/null | python -c "$show" "shard 1/8 $v"; done
