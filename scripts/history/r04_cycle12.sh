#!/bin/bash
# round 4, cycle 12: kernel trace of a many-loci run (2 000 loci x 30 reads through one handle, one host process: the GPU side of bench.py's many_loci)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04_many_loci_prof -o p -- python3 $R/scripts/exp_many_loci.py 2000 4 > $O/r04_many_loci_prof.log 2>&1 || { tail -5 $O/r04_many_loci_prof.log; exit 1; }
grep "batched\|loop" $O/r04_many_loci_prof.log | cut -c1-300
head -14 $O/r04_many_loci_prof/p_kernel_stats.csv | cut -c1-160
