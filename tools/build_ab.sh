#!/bin/bash
# A/B libraries for one GPU session: build_exp/<name>/libmiwave.so (+ the host library beside it) from the working tree
# with extra -D switches; a run picks one with MIWAVE_LIB_DIR=build_exp/<name>. build_exp/ is git-ignored and travels
# to the GPU box. Usage: tools/build_ab.sh <name> [-DMIW_X=1 ...]       (name "head": the kernels of HEAD, from a scratch checkout)
set -e
name=$1; shift
root=$(cd $(dirname $0)/.. && pwd)
src=$root
if [ "$name" = head ]; then
  src=/tmp/miw_head; rm -rf $src; mkdir -p $src
  git -C $root archive HEAD mitsuba2_amd include | tar -x -C $src
fi
out=$root/build_exp/$name; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-gpu-flush-denormals-to-zero -fPIC -shared "$@" $src/mitsuba2_amd/csrc/miwave.hip -o $out/libmiwave.so
g++ -O2 -std=c++17 -ffp-contract=off -mfma -fPIC -shared $src/mitsuba2_amd/host/miwave_host.cpp -o $out/libmiwave_host.so -L$out -lmiwave '-Wl,-rpath,$ORIGIN'
ls -la $out
