#!/bin/bash
# streams x chunks sweep of pipelined calls at small per-GPU shares.  Usage: scripts/stream_share_sweep.sh TAG "12500 25000" "4:4 8:4 ..."
TAG=$1; R=$GRAFT_REPO_ROOT; L=$R/gpurun_out/${TAG}_stream_share_sweep.log
: > $L
for n in $2; do
  steps=$((1000000 / n))
  for sc in $3; do
    s=${sc%%:*}; rest=${sc#*:}; c=${rest%%:*}; q=${rest#*:}; [ "$q" = "$rest" ] && q=8
    out=$(GPU_MAX_HW_QUEUES=$q WSX_STREAMS=$s WSX_CHUNKS=$c timeout -k 10 200 python $R/bench.py --no-cpu-baseline --reads $n --steps $steps --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step  %.4g reads/s' % (d['ms_per_step'], d['value']))") || exit 1
    echo "reads $n streams $s chunks $c hwq $q : $out" | tee -a $L
  done
done
