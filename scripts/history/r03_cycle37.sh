#!/bin/bash
# round 3, cycle 37: kernel trace of the from-raw leg (loader kernels + caller with sequences)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03c37; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export WARPSTR_BENCH_PROFILING=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o p -- python3 $R/bench.py --no-cpu-baseline --no-secondary --from-raw --steps 10 > $O/bench.json 2> $O/trace.err || { tail -5 $O/trace.err; exit 1; }
python3 - <<PY
import csv,glob
f=glob.glob('$O/trace/**/p_kernel_stats.csv',recursive=True)[0]
for r in list(csv.reader(open(f)))[:20]: print(r[0][:70].ljust(70), r[1], r[3][:10], r[4])
PY
