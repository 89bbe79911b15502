#!/bin/bash
# round 3, cycle 13: what a stage costs the pipelined step: the step with that stage's kernel left out (results are garbage)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
export WARPSTR_BENCH_PROFILING=1 WARPSTR_HIP_LIB=$R/build/exp/libskip.so
for rep in 1 2; do for skip in 0 4 8 16 32 56 60; do
  WSX_EXP_SKIP=$skip timeout -k 10 300 python bench.py --no-cpu-baseline --no-verify > $O/r03c13_b.json 2> $O/r03c13_b.err || { tail $O/r03c13_b.err; exit 1; }
  python3 -c "import json; d=json.load(open('$O/r03c13_b.json')); print('skip=$skip', round(d['ms_per_step'],3), 'fill union', round(d['roofline']['fill_union_ms_per_step'],3))"
done; done
