#!/bin/bash
# Sanitizer tier (the reference's MTS_SANITIZE_ADDRESS / MTS_SANITIZE_MEMORY, CMakeLists.txt:80-207). GPU AddressSanitizer
# is not available on this pool, so the sanitizers run where they can: on the CPU build of everything the device and the checker SHARE —
# the leaf headers csrc/miw/*.h, the tree builders and collapses, film_classes.h, and the device STAGES themselves as the CPU wavefront
# emulator runs them (oracle/wavefront_emu.cpp: walk4 / walk8 bodies in columns of exactly `depth` entries, the interleaved log, the
# film replay's gather) — under AddressSanitizer + UndefinedBehaviorSanitizer, driven by the CPU tests that exercise those paths.
#   bash tools/sanitize_cpu.sh [pytest -k expression]        -> profiles/r06_sanitizers.txt (when run from the repo root)
set -e
cd "$(dirname "$0")/.." && root=$(pwd)
out=/tmp/miw_asan; mkdir -p $out
flags="-O1 -g -std=c++17 -ffp-contract=off -mfma -fPIC -shared -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer"
g++ $flags oracle/miw_oracle.cpp oracle/wavefront_emu.cpp -o $out/libmiw_oracle.so -lpthread
g++ $flags -DMIW_SPECTRAL=1 oracle/miw_oracle.cpp oracle/wavefront_emu.cpp -o $out/libmiw_oracle_spectral.so -lpthread
asan=$(gcc -print-file-name=libasan.so); ubsan=$(gcc -print-file-name=libubsan.so)
k=${1:-"kat or emu or walk or bvh or film or spiral or leaves or chunk or oracle or spectral or hier2d or texture or sphere or rect"}
# (python itself is not instrumented: leak reports of the interpreter are noise -> detect_leaks=0; everything else is fatal)
MIW_TEST_NO_BUILD=1 MIW_ORACLE_DIR=$out LD_PRELOAD="$asan $ubsan" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:strict_string_checks=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest tests -x -q -m "not gpu" -k "$k" -p no:cacheprovider 2>&1 | tail -15
