#!/bin/bash
# round 4, cycle 6: per-kernel times of a single-stream step with the generated fill
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export WARPSTR_CACHE_DIR=$O/fillgen_cache
cd /tmp && export TMPDIR=/tmp
WARPSTR_BENCH_PROFILING=1 WSX_STREAMS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04c6_prof -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/r04c6_prof.log 2>&1
python3 $R/scripts/kstats.py $O/r04c6_prof/p_kernel_trace.csv 5
