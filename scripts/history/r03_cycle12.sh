#!/bin/bash
# round 3, cycle 12: tie counters of the flank localisation -- GPU tests of the flank path, tie rate on upstream-shaped input
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_flanks.py tests/test_cabi_exports.py -q -x > $O/r03c12_gpu_tests.log 2>&1 || { tail -60 $O/r03c12_gpu_tests.log; exit 1; }
tail -2 $O/r03c12_gpu_tests.log
timeout -k 10 900 python scripts/exp_flank_ties.py 4000 400 > $O/r03c12_flank_ties.log 2>&1 || { tail -20 $O/r03c12_flank_ties.log; exit 1; }
cat $O/r03c12_flank_ties.log
