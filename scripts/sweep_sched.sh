#!/bin/bash
# Sweep knobs of the batch call on the headline workload (bench.py, no CPU baseline).  Each line: label : ms/step reads/s
R=$GRAFT_REPO_ROOT
run() { # label, env...
  local label="$1"; shift
  local out=$(env "$@" timeout -k 10 120 python $R/bench.py --no-cpu-baseline --steps 5 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step  %.3g reads/s' % (d['ms_per_step'], d['value']))")
  echo "$label : $out"
}
run "default (4 streams, 8 chunks)" X=1
for sc in "4 4" "2 4" "3 3" "4 5" "4 6" "2 2" "3 6" "4 4" "4 3" "4 2"; do set -- $sc
  run "streams=$1 chunks=$2" WSX_STREAMS=$1 WSX_CHUNKS=$2
done
run "default (4 streams, 8 chunks)" X=1
