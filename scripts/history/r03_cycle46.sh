#!/bin/bash
# round 3, cycle 46: fill blocks per CU capped (wave slots left free for the stages of other chunks), headline step, one box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
export WARPSTR_BENCH_PROFILING=1
for rep in 1 2; do for env in "WSX_UNUSED=1" "WSX_FILL_BLOCKS_PER_CU=7" "WSX_FILL_BLOCKS_PER_CU=6" "WSX_FILL_BLOCKS_PER_CU=5"; do
  env $env timeout -k 10 300 python bench.py --no-cpu-baseline > $O/r03c46_b.json 2> $O/r03c46_b.err || { tail $O/r03c46_b.err; exit 1; }
  python3 -c "import json; d=json.load(open('$O/r03c46_b.json')); print('[$env]', round(d['value']), round(d['ms_per_step'],3), 'fill union', round(d['roofline'].get('fill_union_ms_per_step',0),3))"
done; done
