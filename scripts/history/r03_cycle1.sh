#!/bin/bash
# round 3, first cycle: the slot-per-wave fill (dtw_fill_wg) -- parity suite, then cfg1 / cfg5 with the kernel off / on
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
export WARPSTR_BENCH_PROFILING=1
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/r03c1_gpu_tests.log 2>&1 || { tail -60 $O/r03c1_gpu_tests.log; }
tail -2 $O/r03c1_gpu_tests.log
export WARPSTR_BENCH_PROFILING=1
for w in cfg1 cfg5; do for mode in 0 1 2 1 0; do
  WSX_FILL_WG=$mode timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > $O/r03c1_bench_${w}_$mode.json 2> $O/r03c1_bench_${w}_$mode.err || { tail $O/r03c1_bench_${w}_$mode.err; exit 1; }
  python3 -c "import json; d=json.load(open('$O/r03c1_bench_${w}_$mode.json')); print('$w wg=$mode', round(d['value']), round(d['ms_per_step'],3), d['roofline']['kernels'], 'fill union', round(d['roofline'].get('fill_union_ms_per_step',0),3), d.get('verified'))"
done; done
