#!/bin/bash
# round 4, cycle 4: (a) PMC counters of the example loci's fills with bank-aware lanes, (b) the headline under workspace limits,
# (c) the cost of a slot at other read lengths (dist.SLOT_COST)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
scripts/pmc_real_loci.sh r04_real > $O/r04_real_loci_pmc.log 2>&1 || { tail -5 $O/r04_real_loci_pmc.log; exit 1; }
tail -12 $O/r04_real_loci_pmc.log
: > $O/r04_wslimit_sweep.log
for rep in 1 2; do for lim in 0 16 32 64 96; do
  WARPSTR_BENCH_PROFILING=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --workspace-limit-gib $lim 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); w=d['workspace']
print('limit %s GiB: %.3f ms/step  %.4g reads/s  chunks/call %.0f  allocated %.2f GB  %.1f B/sample  (limit in effect %.1f GiB)' % ('$lim' if '$lim'!='0' else 'default', d['ms_per_step'], d['value'], d['config']['chunk_plan']['chunks_per_call'], w['bytes_allocated']/1e9, w['bytes_per_sample'], w['limit_bytes']/2**30))" | tee -a $O/r04_wslimit_sweep.log
done; done
for T in 1000 3000 5000; do
  timeout -k 10 250 python scripts/exp_staircase.py 20000 $T 63,127,191,255,319 2>&1 | grep -v amdgpu.ids
done | tee $O/r04_staircase_T.log
