#!/bin/bash
# round 3, cycle 30: three-candidate automata share the four-candidate kernel when both exist (fewer launch groups), cfg5 + headline
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
export WARPSTR_BENCH_PROFILING=1
line() { python3 -c "import json,sys; d=json.load(open('$1')); print('$2', round(d['value']), round(d['ms_per_step'],3), 'fill union', round(d['roofline'].get('fill_union_ms_per_step',0),3), d.get('verified',{}).get('mismatches'))"; }
for rep in 1 2 3; do for env in "WSX_NO_VARIANT_MERGE=1" "WSX_UNUSED=1"; do
  env $env timeout -k 10 300 python bench.py --workload cfg5 --no-cpu-baseline > $O/r03c30_b.json 2> $O/r03c30_b.err || { tail $O/r03c30_b.err; exit 1; }
  line $O/r03c30_b.json "cfg5 [$env]"
done; done
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/r03c30_tests.log 2>&1; tail -3 $O/r03c30_tests.log
