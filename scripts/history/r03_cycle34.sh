#!/bin/bash
# round 3, cycle 34: streams x chunks per call on the headline step with the round's last library
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
export WARPSTR_BENCH_PROFILING=1
line() { python3 -c "import json,sys; d=json.load(open('$1')); print('$2', round(d['value']), round(d['ms_per_step'],3), 'fill union', round(d['roofline'].get('fill_union_ms_per_step',0),3), d.get('verified',{}).get('mismatches'))"; }
for env in "WSX_UNUSED=1" "WSX_STREAMS=5 WSX_STREAMS_PER_CALL=5 WSX_CHUNKS=5" "WSX_STREAMS=6 WSX_STREAMS_PER_CALL=6 WSX_CHUNKS=6" "WSX_STREAMS=8 WSX_STREAMS_PER_CALL=8 WSX_CHUNKS=8" "WSX_STREAMS=4 WSX_STREAMS_PER_CALL=4 WSX_CHUNKS=8" "WSX_STREAMS=3 WSX_STREAMS_PER_CALL=3 WSX_CHUNKS=3" "WSX_STREAMS=4 WSX_STREAMS_PER_CALL=4 WSX_CHUNKS=6" "WSX_UNUSED=2"; do
  env $env timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary > $O/r03c34_b.json 2> $O/r03c34_b.err || { tail $O/r03c34_b.err; exit 1; }
  line $O/r03c34_b.json "headline [$env]"
done
