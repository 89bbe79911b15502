#!/bin/bash
# Experiment build of the several-slot fills: only the flank-110 variants (K = 4, F = 2; -DWSX_ONLY_WG), extra -D flags
# -> build/exp/lib<NAME>.so   Usage: build_exp_wg.sh NAME [flags...]
set -e
NAME=$1; shift
ROOT=$(cd $(dirname $0)/.. && pwd)
OUT=$ROOT/build/exp; mkdir -p $OUT/$NAME
FLAGS="-O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -fno-gpu-rdc"
/opt/rocm/bin/hipcc $FLAGS -DWSX_ONLY_WG "$@" -c $ROOT/warpstr_amd/csrc/dtw_kernels.hip -o $OUT/$NAME/dtw_kernels.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib$NAME.so $OUT/$NAME/dtw_kernels.o $ROOT/warpstr_amd/csrc/wsx_api.o $ROOT/warpstr_amd/csrc/mid_kernels.o $ROOT/warpstr_amd/csrc/wsx_prep.o $ROOT/warpstr_amd/csrc/flank_kernels.o
echo built $OUT/lib$NAME.so
