#!/bin/bash
# round 3, cycle 6: wave-per-read segmentation -- GPU suite, fuzz, headline / cfg1 A/B against the block-per-read kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/r03c6_gpu_tests.log 2>&1 || { tail -60 $O/r03c6_gpu_tests.log; exit 1; }
tail -2 $O/r03c6_gpu_tests.log
timeout -k 10 600 python scripts/fuzz_parity.py 6 > $O/r03c6_fuzz_parity.log 2>&1 || { tail -30 $O/r03c6_fuzz_parity.log; exit 1; }
tail -4 $O/r03c6_fuzz_parity.log
line() { python3 -c "import json,sys; d=json.load(open('$1')); print('$2', round(d['value']), round(d['ms_per_step'],3), 'fill union', round(d['roofline'].get('fill_union_ms_per_step',0),3), d.get('verified',{}).get('mismatches'))"; }
for rep in 1 2 3; do for env in "WSX_SEGMENT_BLOCK_KERNEL=1" "WSX_X=0"; do
  env $env timeout -k 10 300 python bench.py --no-cpu-baseline > $O/r03c6_b.json 2> $O/r03c6_b.err || { tail $O/r03c6_b.err; exit 1; }
  line $O/r03c6_b.json "headline [$env]"
done; done
for env in "WSX_SEGMENT_BLOCK_KERNEL=1" "WSX_X=0" "WSX_SEGMENT_BLOCK_KERNEL=1" "WSX_X=0"; do
  env $env timeout -k 10 300 python bench.py --workload cfg1 --no-cpu-baseline > $O/r03c6_b.json 2> $O/r03c6_b.err || { tail $O/r03c6_b.err; exit 1; }
  line $O/r03c6_b.json "cfg1 [$env]"
done
