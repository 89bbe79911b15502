#!/bin/bash
# round 3, cycle 14: traceback with a static slot select and the lane-major predecessor shortcut
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
export WARPSTR_BENCH_PROFILING=1
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/r03c14_gpu_tests.log 2>&1 || { tail -60 $O/r03c14_gpu_tests.log; exit 1; }
tail -2 $O/r03c14_gpu_tests.log
timeout -k 10 600 python scripts/fuzz_loci.py 800 48 > $O/r03c14_fuzz_loci.log 2>&1 || { tail -30 $O/r03c14_fuzz_loci.log; exit 1; }
tail -1 $O/r03c14_fuzz_loci.log
line() { python3 -c "import json,sys; d=json.load(open('$1')); print('$2', round(d['value']), round(d['ms_per_step'],3), 'fill union', round(d['roofline'].get('fill_union_ms_per_step',0),3), d.get('verified',{}).get('mismatches'))"; }
for w in cfg1 cfg5 cfg1 cfg5 cfg1; do
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > $O/r03c14_b.json 2> $O/r03c14_b.err || { tail $O/r03c14_b.err; exit 1; }
  line $O/r03c14_b.json "$w"
done
timeout -k 10 300 python scripts/exp_staircase.py > $O/r03c14_stair.log 2>&1 || { tail $O/r03c14_stair.log; exit 1; }
grep "S = " $O/r03c14_stair.log
