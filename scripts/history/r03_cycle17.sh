#!/bin/bash
# round 3, cycle 17: reads per workgroup of the lane-major fill (1 / 2 / 4), cfg1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
export WARPSTR_BENCH_PROFILING=1
line() { python3 -c "import json,sys; d=json.load(open('$1')); print('$2', round(d['value']), round(d['ms_per_step'],3), 'fill union', round(d['roofline'].get('fill_union_ms_per_step',0),3))"; }
for rep in 1 2 3; do for lib in wpb4 wpb2 wpb1; do
  WARPSTR_HIP_LIB=$R/build/exp/lib$lib.so timeout -k 10 300 python bench.py --workload cfg1 --no-cpu-baseline --no-verify > $O/r03c17_b.json 2> $O/r03c17_b.err || { tail $O/r03c17_b.err; exit 1; }
  line $O/r03c17_b.json "cfg1 [$lib]"
done; done
