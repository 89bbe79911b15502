#!/bin/bash
# rocprofv3 kernel statistics of the from_fast5 leg in one process (no child processes under the profiler): the VBZ decoder beside
# the signal loader and the caller; then the parity sweeps on the round's last library.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
WARPSTR_BENCH_FAST5_ONLY=one_process rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fast5 -o fast5 -- python $R/scripts/exp_from_fast5.py 600 > $O/r05_from_fast5_under_trace.json 2> $O/r05_from_fast5_under_trace.err
find $O/prof_fast5 -name '*kernel_stats.csv' -exec cp {} $O/r05_from_fast5_kernel_stats.csv \;
head -12 $O/r05_from_fast5_kernel_stats.csv | cut -c1-160
cd $R
python scripts/fuzz_loci.py 600 48 > $O/r05_fuzz_loci_600_final.log 2>&1; tail -2 $O/r05_fuzz_loci_600_final.log
python scripts/fuzz_parity.py 6 > $O/r05_fuzz_parity_final.log 2>&1; tail -2 $O/r05_fuzz_parity_final.log
