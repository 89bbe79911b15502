#!/bin/bash
# round 4, cycle 3: the example loci at flank 110 with bank-aware lanes; the stacked layout at four slots against the slot-major kernel (experiment build)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export WARPSTR_HIP_LIB=$R/build/exp/libr04exp.so
for rep in 1 2; do
for env in "WSX_STACKED_MIN_K=5" "WSX_STACKED_MIN_K=4"; do
  echo "[$env]"; env $env timeout -k 10 300 python scripts/exp_real_loci.py 2>&1 | grep -v amdgpu.ids | grep "HD\|DM2\|AAAT"
done; done | tee $O/r04c3_real_loci_ab.log
