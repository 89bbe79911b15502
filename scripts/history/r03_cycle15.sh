#!/bin/bash
# round 3, cycle 15: cfg1 chunk / stream sweep; slot-per-wave vs one-wave kernel on one box (192 / 256 states)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
export WARPSTR_BENCH_PROFILING=1
line() { python3 -c "import json,sys; d=json.load(open('$1')); print('$2', round(d['value']), round(d['ms_per_step'],3), 'fill union', round(d['roofline'].get('fill_union_ms_per_step',0),3), d['roofline']['launches_per_step'])"; }
for env in "WSX_X=0" "WSX_CHUNKS=2" "WSX_CHUNKS=3" "WSX_CHUNKS=4" "WSX_CHUNKS=6" "WSX_CHUNKS=8" "WSX_X=0" "WSX_CHUNKS=4"; do
  env $env timeout -k 10 300 python bench.py --workload cfg1 --no-cpu-baseline --no-verify > $O/r03c15_b.json 2> $O/r03c15_b.err || { tail $O/r03c15_b.err; exit 1; }
  line $O/r03c15_b.json "cfg1 [$env]"
done
for rep in 1 2; do for env in "WSX_FILL_WG=0" "WSX_FILL_WG=1"; do
  env $env timeout -k 10 300 python scripts/exp_staircase.py 20000 2000 192,256 > $O/r03c15_stair.log 2>&1 || { tail $O/r03c15_stair.log; exit 1; }
  echo "[$env]"; grep "S = " $O/r03c15_stair.log
done; done
