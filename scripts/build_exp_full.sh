#!/bin/bash
# Experiment build of the WHOLE library (every fill variant) with -DWSX_EXPERIMENT: the A/B switches of scripts/ are read from
# the environment by such builds only (wsx_device.h: wsx_exp_env).  -> build/exp/lib<NAME>.so   Usage: build_exp_full.sh NAME [flags...]
set -e
NAME=$1; shift
ROOT=$(cd $(dirname $0)/.. && pwd)
OUT=$ROOT/build/exp/$NAME; mkdir -p $OUT
FLAGS="-O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -fno-gpu-rdc -DWSX_EXPERIMENT $*"
for f in wsx_api dtw_kernels mid_kernels wsx_prep flank_kernels; do
  /opt/rocm/bin/hipcc $FLAGS -c $ROOT/warpstr_amd/csrc/$f.hip -o $OUT/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build/exp/lib$NAME.so $OUT/wsx_api.o $OUT/dtw_kernels.o $OUT/mid_kernels.o $OUT/wsx_prep.o $OUT/flank_kernels.o
echo built $ROOT/build/exp/lib$NAME.so
