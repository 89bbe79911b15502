#!/bin/bash
# chunks-per-call sweep at small per-GPU shares.  Usage: scripts/chunk_share_sweep.sh TAG
TAG=$1; R=$GRAFT_REPO_ROOT; L=$R/gpurun_out/${TAG}_chunk_share_sweep.log
: > $L
for n in 12500 25000 50000; do
  steps=$((1000000 / n))
  for c in 1 2 3 4 6 8; do
    out=$(WSX_CHUNKS=$c timeout -k 10 200 python $R/bench.py --no-cpu-baseline --reads $n --steps $steps --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step  %.4g reads/s' % (d['ms_per_step'], d['value']))") || exit 1
    echo "reads $n chunks $c : $out" | tee -a $L
  done
done
