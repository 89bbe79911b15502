#!/bin/bash
# round 3, cycle 24: lane-major with slots 0, 1 and K-1 exporting (LM = 3): parity, then the staircase around 257 states
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
export WARPSTR_BENCH_PROFILING=1
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/r03c24_gpu_tests.log 2>&1 || { tail -60 $O/r03c24_gpu_tests.log; exit 1; }
tail -1 $O/r03c24_gpu_tests.log
timeout -k 10 600 python scripts/fuzz_loci.py 1000 48 > $O/r03c24_fuzz_loci.log 2>&1 || { tail -30 $O/r03c24_fuzz_loci.log; exit 1; }
grep ", 3>" $O/r03c24_fuzz_loci.log; tail -1 $O/r03c24_fuzz_loci.log
for rep in 1 2; do for env in "WSX_FILL_LM=2" "WSX_FILL_LM=1"; do
  env $env timeout -k 10 300 python scripts/exp_staircase.py 20000 2000 257 > $O/r03c24_stair.log 2>&1 || { tail $O/r03c24_stair.log; exit 1; }
  echo "[$env]"; grep "S = " $O/r03c24_stair.log
done; done
