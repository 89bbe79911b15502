#!/bin/bash
# round 3, cycle 32: loci fuzz with rescaling.threshold > 1 and perturbed levels (FITPACK's smoothing branch on the device)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout -k 10 1000 python scripts/fuzz_loci.py 400 32 --smooth > $O/r03c32_fuzz_smooth.log 2>&1; rc=$?; tail -25 $O/r03c32_fuzz_smooth.log; exit $rc
