set -x
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "film_replay or float32_film or ragged or cornell_diffuse or golden_films or tiny_blocks" 2>&1 | tail -5
export MIW_BENCH_NO_LIVE=1
for q in 24 42 0 28 44 24 42 0; do MIW_FILM_QUADS=$q timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('quads=$q', round(d['value'],1), round(d['ms_per_step'],2), d['roofline']['kernel_ms'])"; done
