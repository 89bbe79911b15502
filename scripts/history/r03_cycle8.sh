#!/bin/bash
# round 3, cycle 8: wave-per-read segmentation with its list in global scratch: parity on the mask tests, headline A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/r03c8_gpu_tests.log 2>&1 || { tail -60 $O/r03c8_gpu_tests.log; exit 1; }
tail -2 $O/r03c8_gpu_tests.log
line() { python3 -c "import json,sys; d=json.load(open('$1')); print('$2', round(d['value']), round(d['ms_per_step'],3), 'fill union', round(d['roofline'].get('fill_union_ms_per_step',0),3), d.get('verified',{}).get('mismatches'))"; }
for rep in 1 2 3; do for env in "WSX_SEGMENT_BLOCK_KERNEL=1" "WSX_X=0"; do
  env $env timeout -k 10 300 python bench.py --no-cpu-baseline > $O/r03c8_b.json 2> $O/r03c8_b.err || { tail $O/r03c8_b.err; exit 1; }
  line $O/r03c8_b.json "headline [$env]"
done; done
