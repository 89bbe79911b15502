#!/bin/bash
# round 3, cycle 21: low-priority streams for the stages between the fills x more chunks in flight
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
export WARPSTR_BENCH_PROFILING=1 GPU_MAX_HW_QUEUES=16
line() { python3 -c "import json,sys; d=json.load(open('$1')); print('$2', round(d['value']), round(d['ms_per_step'],3), 'fill union', round(d['roofline'].get('fill_union_ms_per_step',0),3), d.get('verified',{}).get('mismatches'))"; }
for env in "WSX_MID_STREAMS=0" "WSX_MID_STREAMS=1 WSX_CHUNKS=8" "WSX_MID_STREAMS=1 WSX_STREAMS=8 WSX_STREAMS_PER_CALL=8 WSX_CHUNKS=8" "WSX_MID_STREAMS=1 WSX_STREAMS=8 WSX_STREAMS_PER_CALL=8 WSX_CHUNKS=16" "WSX_MID_STREAMS=0 WSX_STREAMS=8 WSX_STREAMS_PER_CALL=8 WSX_CHUNKS=8" "WSX_MID_STREAMS=1 WSX_STREAMS=6 WSX_STREAMS_PER_CALL=6 WSX_CHUNKS=6" "WSX_MID_STREAMS=1 WSX_INFLIGHT=4"; do
  for w in headline; do
    env $env timeout -k 10 300 python bench.py --no-cpu-baseline > $O/r03c21_b.json 2> $O/r03c21_b.err || { tail $O/r03c21_b.err; exit 1; }
    line $O/r03c21_b.json "$w [$env]"
  done
done
