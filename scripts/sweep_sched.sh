#!/bin/bash
# Sweep knobs of the batch call on the headline workload (bench.py, no CPU baseline).  Each line: label : ms/step reads/s
R=$GRAFT_REPO_ROOT
run() { # label, env...
  local label="$1"; shift
  local out=$(env "$@" timeout -k 10 120 python $R/bench.py --no-cpu-baseline --steps 5 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step  %.3g reads/s' % (d['ms_per_step'], d['value']))")
  echo "$label : $out"
}
run "default" X=1
for sc in "4 4" "5 5" "6 6" "8 8" "6 12" "8 16" "3 3" "4 8"; do set -- $sc
  run "streams=$1 chunks=$2" WSX_STREAMS=$1 WSX_CHUNKS=$2 GPU_MAX_HW_QUEUES=16
done
run "default" X=1
