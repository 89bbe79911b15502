#!/bin/bash
# pipelined small calls: chunks per call x calls in flight on 4 streams.  Usage: scripts/small_call_sweep.sh TAG
TAG=$1; R=$GRAFT_REPO_ROOT; L=$R/gpurun_out/${TAG}_small_call_sweep.log; : > $L
export WARPSTR_BENCH_PROFILING=1 WSX_STREAMS=4
for n in 12500 25000 50000; do
  steps=$((1000000 / n))
  for cfg in "1 2" "1 3" "1 4" "2 2" "2 4" "4 2" "4 4"; do
    set -- $cfg
    out=$(WSX_CHUNKS=$1 WSX_INFLIGHT=$2 timeout -k 10 200 python $R/bench.py --no-cpu-baseline --no-verify --reads $n --steps $steps --warmup 4 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step  %.4g reads/s' % (d['ms_per_step'], d['value']))") || exit 1
    echo "reads $n chunks $1 in_flight $2 : $out" | tee -a $L
  done
done
