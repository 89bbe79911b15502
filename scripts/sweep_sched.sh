#!/bin/bash
# Sweep knobs of the batch call on the headline workload (bench.py, no CPU baseline).  Each line: label : ms/step reads/s
R=$GRAFT_REPO_ROOT
run() { # label, env...
  local label="$1"; shift
  local out=$(env "$@" timeout -k 10 120 python $R/bench.py --no-cpu-baseline --steps 5 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step  %.3g reads/s' % (d['ms_per_step'], d['value']))")
  echo "$label : $out"
}
for rep in 1 2; do
run "stream tb, chunks=8" X=1
run "stream tb, chunks=4" WSX_CHUNKS=4
run "wave tb, chunks=8" WSX_STREAM_TRACEBACK_MIN=100000000
run "wave tb, chunks=4" WSX_STREAM_TRACEBACK_MIN=100000000 WSX_CHUNKS=4
run "wave tb, chunks=6 streams=3" WSX_STREAM_TRACEBACK_MIN=100000000 WSX_CHUNKS=6 WSX_STREAMS=3
run "wave tb, chunks=16" WSX_STREAM_TRACEBACK_MIN=100000000 WSX_CHUNKS=16
done
