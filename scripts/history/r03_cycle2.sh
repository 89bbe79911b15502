#!/bin/bash
# round 3, cycle 2: lane-major fill -- parity suite with it on, cfg1 A/B over WSX_FILL_LM / WSX_FILL_WG and register budgets
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
export WARPSTR_BENCH_PROFILING=1
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/r03c2_gpu_tests.log 2>&1 || { tail -60 $O/r03c2_gpu_tests.log; }
tail -2 $O/r03c2_gpu_tests.log
line() { python3 -c "import json,sys; d=json.load(open('$1')); print('$2', round(d['value']), round(d['ms_per_step'],3), d['roofline']['kernels'], 'fill union', round(d['roofline'].get('fill_union_ms_per_step',0),3), d.get('verified',{}).get('mismatches'))"; }
for rep in 1 2; do
for env in "WSX_FILL_LM=0 WSX_FILL_WG=0" "WSX_FILL_LM=0 WSX_FILL_WG=1" "WSX_FILL_LM=1" "WSX_FILL_LM=2"; do
  env $env timeout -k 10 300 python bench.py --workload cfg1 --no-cpu-baseline > $O/r03c2_b.json 2> $O/r03c2_b.err || { tail $O/r03c2_b.err; exit 1; }
  line $O/r03c2_b.json "cfg1 [$env]"
done
for lib in lmw0 lmw4 lmw5 lmw6; do
  WARPSTR_HIP_LIB=$R/build/exp/lib$lib.so timeout -k 10 300 python bench.py --workload cfg1 --no-cpu-baseline > $O/r03c2_b.json 2> $O/r03c2_b.err || { tail $O/r03c2_b.err; exit 1; }
  line $O/r03c2_b.json "cfg1 [$lib]"
done
done
