#!/bin/bash
# round 3, cycle 9: corner cut forced for M rows only -- GPU suite, random loci, cfg1 / headline / staircase
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
export WARPSTR_BENCH_PROFILING=1
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/r03c9_gpu_tests.log 2>&1 || { tail -60 $O/r03c9_gpu_tests.log; exit 1; }
tail -2 $O/r03c9_gpu_tests.log
timeout -k 10 600 python scripts/fuzz_loci.py 800 48 > $O/r03c9_fuzz_loci.log 2>&1 || { tail -30 $O/r03c9_fuzz_loci.log; exit 1; }
tail -1 $O/r03c9_fuzz_loci.log
line() { python3 -c "import json,sys; d=json.load(open('$1')); print('$2', round(d['value']), round(d['ms_per_step'],3), 'fill union', round(d['roofline'].get('fill_union_ms_per_step',0),3), d.get('verified',{}).get('mismatches'))"; }
for w in cfg1 headline cfg5 cfg1 headline; do
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > $O/r03c9_b.json 2> $O/r03c9_b.err || { tail $O/r03c9_b.err; exit 1; }
  line $O/r03c9_b.json "$w"
done
timeout -k 10 300 python scripts/exp_staircase.py > $O/r03c9_stair.log 2>&1 || { tail $O/r03c9_stair.log; exit 1; }
grep "S = " $O/r03c9_stair.log
