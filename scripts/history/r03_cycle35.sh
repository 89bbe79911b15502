#!/bin/bash
# round 3, cycle 35: stacked lane-major placement (LM = 4): parity on random loci, the example loci at flank 110
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout -k 10 300 python scripts/exp_real_loci.py > $O/r03c35_real_loci.log 2>&1 || { tail -5 $O/r03c35_real_loci.log; exit 1; }
cat $O/r03c35_real_loci.log
WSX_NO_STACKED=1 timeout -k 10 300 python scripts/exp_real_loci.py > $O/r03c35_real_loci_off.log 2>&1 || { tail -5 $O/r03c35_real_loci_off.log; exit 1; }
cat $O/r03c35_real_loci_off.log
timeout -k 10 500 python scripts/fuzz_loci.py 600 32 > $O/r03c35_fuzz.log 2>&1; rc=$?; grep -c MISMATCH $O/r03c35_fuzz.log; tail -12 $O/r03c35_fuzz.log; exit $rc
