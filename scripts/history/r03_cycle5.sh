#!/bin/bash
# round 3, cycle 5: GPU suite, then the profiles kept under profiles/ (kernel trace + PMC passes) for cfg1 and the headline
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
export WARPSTR_BENCH_PROFILING=1
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/r03c5_gpu_tests.log 2>&1 || { tail -60 $O/r03c5_gpu_tests.log; exit 1; }
tail -2 $O/r03c5_gpu_tests.log
scripts/profile_round.sh r03c5_cfg1 --workload cfg1 && scripts/profile_round.sh r03c5_headline
