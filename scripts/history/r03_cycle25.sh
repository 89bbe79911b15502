#!/bin/bash
# round 3: the state staircase on the round's final library
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout -k 10 600 python scripts/exp_staircase.py > $O/r03_stair_final.log 2>&1 || { tail $O/r03_stair_final.log; exit 1; }
grep "S = \|reads x" $O/r03_stair_final.log
