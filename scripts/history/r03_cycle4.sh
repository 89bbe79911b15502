#!/bin/bash
# round 3, cycle 4: lane-major fill on more shapes: random loci against the oracle, the state staircase, cfg5, headline
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
export WARPSTR_BENCH_PROFILING=1
timeout -k 10 600 python scripts/fuzz_loci.py 1500 48 > $O/r03c4_fuzz_loci.log 2>&1 || { tail -30 $O/r03c4_fuzz_loci.log; exit 1; }
tail -32 $O/r03c4_fuzz_loci.log
for env in "WSX_FILL_LM=0 WSX_FILL_WG=0" "WSX_FILL_LM=1"; do
  echo "== staircase [$env]"
  env $env timeout -k 10 300 python scripts/exp_staircase.py > $O/r03c4_stair.log 2>&1 || { tail $O/r03c4_stair.log; exit 1; }
  grep "S = " $O/r03c4_stair.log
done
line() { python3 -c "import json,sys; d=json.load(open('$1')); print('$2', round(d['value']), round(d['ms_per_step'],3), d['roofline']['kernels'], 'fill union', round(d['roofline'].get('fill_union_ms_per_step',0),3), d.get('verified',{}).get('mismatches'))"; }
for env in "WSX_FILL_LM=0 WSX_FILL_WG=0" "WSX_FILL_LM=1" "WSX_FILL_LM=0 WSX_FILL_WG=0" "WSX_FILL_LM=1"; do
  env $env timeout -k 10 300 python bench.py --workload cfg5 --no-cpu-baseline > $O/r03c4_b.json 2> $O/r03c4_b.err || { tail $O/r03c4_b.err; exit 1; }
  line $O/r03c4_b.json "cfg5 [$env]"
done
