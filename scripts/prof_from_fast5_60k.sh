#!/bin/bash
# The 60 000-read from_fast5 run (reader processes) under a kernel + memory-copy trace: what the GPU does per batch when the readers
# are ahead of it.  Usage (GPU box): scripts/prof_from_fast5_60k.sh [readers]   -> gpurun_out/prof_ff60k/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_ff60k; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export WARPSTR_BENCH_READER_SWEEP=${1:-16} WARPSTR_BENCH_FAST5_ONLY=reader_sweep WARPSTR_BENCH_TIMELINE=1
timeout -k 10 500 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/trace -o ff -- python3 $R/scripts/exp_from_fast5.py 1500 > $O/run.json 2> $O/run.err || { tail -5 $O/run.err; exit 1; }
find $O/trace -name '*_stats.csv' -exec cp {} $O/ \;
find $O/trace -name '*kernel_trace.csv' -exec cp {} $O/kernel_trace.csv \;
find $O/trace -name '*memory_copy_trace.csv' -exec cp {} $O/memory_copy_trace.csv \;
rm -rf $O/trace
ls -la $O
