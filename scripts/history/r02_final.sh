#!/bin/bash
# The measurements DESIGN.md quotes for round 2: bench lines (three workloads, from-raw leg), share sweep, profiles.  Usage: scripts/r02_final.sh TAG
R=$GRAFT_REPO_ROOT; TAG=$1
export WARPSTR_BENCH_PROFILING=1
for w in headline cfg1 cfg5; do scripts/profile_round.sh ${TAG}_$w --workload $w > $R/gpurun_out/${TAG}_profile_$w.log 2>&1 || { tail $R/gpurun_out/${TAG}_profile_$w.log; exit 1; }; done
unset WARPSTR_BENCH_PROFILING
echo profiles done
