#!/bin/bash
# round 3, cycle 31: FITPACK's smoothing branch on the device (rescaling.threshold > 1), parity suite
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/r03c31_tests.log 2>&1; rc=$?; tail -15 $O/r03c31_tests.log; exit $rc
