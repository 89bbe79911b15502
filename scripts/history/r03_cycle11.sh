#!/bin/bash
# round 3, cycle 11: threshold > 1 accepted (fit-smooth status), median3/median5 on the device -- GPU suite
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $O/r03c11_gpu_tests.log 2>&1 || { tail -80 $O/r03c11_gpu_tests.log; exit 1; }
tail -2 $O/r03c11_gpu_tests.log
