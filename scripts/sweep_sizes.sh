#!/bin/bash
R=$GRAFT_REPO_ROOT
run() { local label="$1"; shift
  local out=$(env "$@" timeout -k 10 120 python $R/bench.py --no-cpu-baseline --steps 5 --warmup 1 $ARGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step  %.3g reads/s' % (d['ms_per_step'], d['value']))")
  echo "$ARGS $label : $out"; }
for ARGS in "--reads 100000" "--reads 200000" "--reads 50000" "--reads 150000" "--reads 40000 --samples 5000"; do
for c in 4 8 16; do run "chunks=$c" WSX_CHUNKS=$c; done; run "default" X=1; done
