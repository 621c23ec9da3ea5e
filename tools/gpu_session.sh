#!/bin/bash
# r5p: film replay group shapes (k_film_columns<GW, GH>): 4x2 (default), 2x4, 2x2 on C2
tag=${1:-r5p}; out=gpurun_out; mkdir -p $out
timeout 900 python tools/ab_render.py --scenes cornell:512 --reps 2 --set "" --set MIW_FILM_COLUMNS=24 --set MIW_FILM_COLUMNS=22 --set MIW_FILM_COLUMNS=44 > $out/${tag}.txt 2> $out/${tag}.err; cat $out/${tag}.txt; tail -3 $out/${tag}.err
