#!/bin/bash
# Sweep the scheduling knobs of the batch call on the headline workload (bench.py, no CPU baseline).
R=$GRAFT_REPO_ROOT
run() { # label, env...
  local label="$1"; shift
  local out=$(env "$@" timeout -k 10 120 python $R/bench.py --no-cpu-baseline --steps 5 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step  %.3g reads/s' % (d['ms_per_step'], d['value']))")
  echo "$label : $out"
}
for fs in 0 1; do
  run "fill_streams=$fs default" WSX_FILL_STREAMS=$fs
  run "fill_streams=$fs chunks=4" WSX_FILL_STREAMS=$fs WSX_CHUNKS=4
  run "fill_streams=$fs chunks=6 streams=3" WSX_FILL_STREAMS=$fs WSX_CHUNKS=6 WSX_STREAMS=3
  run "fill_streams=$fs chunks=4 streams=2" WSX_FILL_STREAMS=$fs WSX_CHUNKS=4 WSX_STREAMS=2
  run "fill_streams=$fs chunks=2 streams=2" WSX_FILL_STREAMS=$fs WSX_CHUNKS=2 WSX_STREAMS=2
  run "fill_streams=$fs chunks=8 occ=7" WSX_FILL_STREAMS=$fs WSX_CHUNKS=8 WSX_FILL_BLOCKS_PER_CU=7
  run "fill_streams=$fs chunks=4 occ=7" WSX_FILL_STREAMS=$fs WSX_CHUNKS=4 WSX_FILL_BLOCKS_PER_CU=7
  run "fill_streams=$fs chunks=12 streams=6" WSX_FILL_STREAMS=$fs WSX_CHUNKS=12 WSX_STREAMS=6
done
