#!/bin/bash
# round 3, final library: the whole GPU suite, then the bench lines kept under profiles/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/r03_final_gpu_tests.log 2>&1; rc=$?; tail -3 $O/r03_final_gpu_tests.log
[ $rc -eq 0 ] || exit $rc
bash scripts/r03_lines.sh
