#!/bin/bash
# round 3, cycle 29: one side stream per work set for the minority launch groups, cfg5
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
export WARPSTR_BENCH_PROFILING=1
line() { python3 -c "import json,sys; d=json.load(open('$1')); print('$2', round(d['value']), round(d['ms_per_step'],3), 'fill union', round(d['roofline'].get('fill_union_ms_per_step',0),3), d.get('verified',{}).get('mismatches'))"; }
for rep in 1 2; do for env in "WSX_GROUP_STREAMS=0" "WSX_GROUP_STREAMS=1" "WSX_GROUP_STREAMS=1 GPU_MAX_HW_QUEUES=16"; do
  env $env timeout -k 10 300 python bench.py --workload cfg5 --no-cpu-baseline > $O/r03c29_b.json 2> $O/r03c29_b.err || { tail $O/r03c29_b.err; exit 1; }
  line $O/r03c29_b.json "cfg5 [$env]"
done; done
