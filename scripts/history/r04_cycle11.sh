#!/bin/bash
# round 4, cycle 11: chunks per call x calls in flight on the headline (the back-pointer layout changed this round)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for rep in 1 2; do for ch in 4 6 8 12; do
  WSX_CHUNKS=$ch WARPSTR_BENCH_PROFILING=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-verify 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('chunks $ch: %.3f ms/step  %.4g reads/s' % (d['ms_per_step'], d['value']))"
done; done | tee $O/r04_chunk_sweep.log
