#!/bin/bash
# round 3, cycle 28: launch groups of a chunk on side streams (mixed-locus batches): GPU suite, soak, cfg5 / cfg1 / headline A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
export WARPSTR_BENCH_PROFILING=1
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $O/r03c28_gpu_tests.log 2>&1 || { tail -60 $O/r03c28_gpu_tests.log; exit 1; }
tail -1 $O/r03c28_gpu_tests.log
timeout -k 10 600 python scripts/soak_pipelined.py > $O/r03c28_soak.log 2>&1 || { tail -20 $O/r03c28_soak.log; exit 1; }
tail -1 $O/r03c28_soak.log
line() { python3 -c "import json,sys; d=json.load(open('$1')); print('$2', round(d['value']), round(d['ms_per_step'],3), 'fill union', round(d['roofline'].get('fill_union_ms_per_step',0),3), d.get('verified',{}).get('mismatches'))"; }
for rep in 1 2 3; do for env in "WSX_GROUP_STREAMS=0" "WSX_GROUP_STREAMS=1"; do
  env $env timeout -k 10 300 python bench.py --workload cfg5 --no-cpu-baseline > $O/r03c28_b.json 2> $O/r03c28_b.err || { tail $O/r03c28_b.err; exit 1; }
  line $O/r03c28_b.json "cfg5 [$env]"
done; done
for env in "WSX_GROUP_STREAMS=0" "WSX_GROUP_STREAMS=1"; do
  env $env timeout -k 10 300 python bench.py --no-cpu-baseline > $O/r03c28_b.json 2> $O/r03c28_b.err || { tail $O/r03c28_b.err; exit 1; }
  line $O/r03c28_b.json "headline [$env]"
done
