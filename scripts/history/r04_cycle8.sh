#!/bin/bash
# round 4, cycle 8: generator variants of the generated fill, fills alone
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export WARPSTR_CACHE_DIR=$O/fillgen_cache
for o in "sb=1,wpe=2" "sb=0,wpe=2" "sb=1,wpe=0" "sb=1,wpe=2,prio=0" "sb=1,wpe=1"; do
  WARPSTR_FILLGEN_OPTS=$o timeout -k 10 200 python scripts/exp_genfill.py 2>&1 | grep -v amdgpu.ids | grep "generated\|built-in\|Error" 
done | tee $O/r04c8_genfill_variants.log
for n in 32768 65536 131072; do WARPSTR_FILLGEN_OPTS="sb=1,wpe=2" timeout -k 10 200 python scripts/exp_genfill.py $n 2>&1 | grep "generated\|built-in"; done | tee -a $O/r04c8_genfill_variants.log
