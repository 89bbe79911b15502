#!/bin/bash
# round 3, cycle 40 (experiment): DM2 at flank 110 with both strands in the five-slot stacked kernel (one launch group)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for env in "WSX_UNUSED=1" "WSX_EXP_PROMOTE_K5=1 WSX_EXP_PREFER_STACKED=1" "WSX_EXP_PROMOTE_K5=1" "WSX_UNUSED=2"; do
  echo "[$env]"; env $env timeout -k 10 300 python scripts/exp_real_loci.py 2>&1 | grep -v amdgpu.ids | grep "HD\|DM2\|AAAT"
done
