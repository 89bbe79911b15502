#!/bin/bash
# round 3, cycle 36: stacked placement for five slots only: GPU suite, the example loci, random loci
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/r03c36_tests.log 2>&1; rc=$?; tail -3 $O/r03c36_tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python scripts/exp_real_loci.py > $O/r03c36_real_loci.log 2>&1 || { tail -5 $O/r03c36_real_loci.log; exit 1; }
cat $O/r03c36_real_loci.log
timeout -k 10 500 python scripts/fuzz_loci.py 800 32 > $O/r03c36_fuzz.log 2>&1; rc=$?; tail -3 $O/r03c36_fuzz.log; exit $rc
