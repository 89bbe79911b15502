#!/bin/bash
# Experiment build: headline fill variant only, extra -D flags -> build/exp/lib<NAME>.so   Usage: build_exp.sh NAME [flags...]
set -e
NAME=$1; shift
ROOT=$(cd $(dirname $0)/.. && pwd)
OUT=$ROOT/build/exp; mkdir -p $OUT/$NAME
FLAGS="-O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -fno-gpu-rdc -DWSX_ONLY_DEFAULT $*"
for f in wsx_api dtw_kernels mid_kernels wsx_prep flank_kernels; do
  if [ $f = dtw_kernels ] || [ $f = mid_kernels ] || [ ! -f $OUT/base_$f.o ]; then
    tgt=$OUT/$NAME/$f.o; [ $f != dtw_kernels ] && [ $f != mid_kernels ] && tgt=$OUT/base_$f.o
    /opt/rocm/bin/hipcc $FLAGS -c $ROOT/warpstr_amd/csrc/$f.hip -o $tgt &
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib$NAME.so $OUT/$NAME/dtw_kernels.o $OUT/base_wsx_api.o $OUT/$NAME/mid_kernels.o $OUT/base_wsx_prep.o $OUT/base_flank_kernels.o
echo built $OUT/lib$NAME.so
