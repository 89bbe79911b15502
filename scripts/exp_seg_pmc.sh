#!/bin/bash
# Counters of segment_kernel for two library builds.  Usage: scripts/exp_seg_pmc.sh "libA.so libB.so"
R=$GRAFT_REPO_ROOT
for lib in $1; do
  echo "== $lib"
  WARPSTR_HIP_LIB=$R/$lib WARPSTR_BENCH_PROFILING=1 WSX_STREAMS=1 WSX_CHUNKS=1 $R/scripts/pmc_kernel.sh segpmc_$(basename $lib .so) segment_kernel -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 | grep -E "GRBM_GUI|SQ_WAVES|SQ_INSTS_VALU|SQ_INSTS_SALU|SQ_INSTS_LDS|SQ_WAIT_INST_ANY|SQ_ACTIVE_INST_ANY|SQ_WAVE_CYCLES|SQ_BUSY|SQ_INSTS_VMEM|SQ_INSTS_BRANCH|FETCH|WRITE"
done
