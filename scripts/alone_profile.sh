#!/bin/bash
# Per-kernel durations when every kernel runs alone (one stream, one chunk), at several launch sizes.
# Usage: scripts/alone_profile.sh TAG "3125 12500"
TAG=$1; R=$GRAFT_REPO_ROOT; L=$R/gpurun_out/${TAG}_alone.log
: > $L
cd /tmp && export TMPDIR=/tmp
for n in $2; do
  WSX_STREAMS=1 WSX_CHUNKS=1 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_alone_$n -o p -- python3 $R/bench.py --no-cpu-baseline --reads $n --steps 10 --warmup 2 > $R/gpurun_out/${TAG}_alone_$n.log 2>&1 || exit 1
  echo "== reads $n, one stream, one chunk (ms per step; bench has 13 steps incl. warm-up and the extra one)" | tee -a $L
  python3 $R/scripts/kstats.py $R/gpurun_out/${TAG}_alone_$n/p_kernel_trace.csv 13 | tee -a $L
done
