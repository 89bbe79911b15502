#!/bin/bash
# round 3, cycle 33: borders as a wave per read at every launch size (same-box A/B on the headline step)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
export WARPSTR_BENCH_PROFILING=1
line() { python3 -c "import json,sys; d=json.load(open('$1')); print('$2', round(d['value']), round(d['ms_per_step'],3), 'fill union', round(d['roofline'].get('fill_union_ms_per_step',0),3), d.get('verified',{}).get('mismatches'))"; }
for rep in 1 2 3; do for env in "WSX_BORDERS_WAVE_BELOW=8192" "WSX_BORDERS_WAVE_BELOW=100000000"; do
  env $env timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary > $O/r03c33_b.json 2> $O/r03c33_b.err || { tail $O/r03c33_b.err; exit 1; }
  line $O/r03c33_b.json "headline [$env]"
done; done
