#!/bin/bash
# round 3, cycle 19: after splitting run_batch (validate / plan / enqueue): GPU suite + pipelined soak
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $O/r03c19_gpu_tests.log 2>&1 || { tail -60 $O/r03c19_gpu_tests.log; exit 1; }
tail -2 $O/r03c19_gpu_tests.log
timeout -k 10 600 python scripts/soak_pipelined.py > $O/r03c19_soak.log 2>&1 || { tail -20 $O/r03c19_soak.log; exit 1; }
tail -3 $O/r03c19_soak.log
