#!/bin/bash
# A/B of experiment builds (scripts/build_exp_full.sh NAME ...) on one box: the example loci at flank 110 (exp_real_loci.py) and
# configs[4]'s share (bench.py --workload cfg5), each library in turn, twice.   Usage: scripts/ab_libs_real_loci.sh OUT.log NAME...
OUT=$1; shift
R=$GRAFT_REPO_ROOT
: > $OUT
for round in 1 2; do
  for name in "$@"; do
    echo "== $name (round $round)" >> $OUT
    WARPSTR_HIP_LIB=$R/build/exp/lib$name.so WARPSTR_BENCH_PROFILING=1 timeout -k 10 300 python3 $R/scripts/exp_real_loci.py 20000 3000 2>&1 | grep -v "^20000 reads" >> $OUT
    WARPSTR_HIP_LIB=$R/build/exp/lib$name.so WARPSTR_BENCH_PROFILING=1 timeout -k 10 300 python3 $R/bench.py --workload cfg5 --no-cpu-baseline 2> /dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cfg5  %.2f ms per step  %.3f M reads/s  verified %s' % (d['ms_per_step'], d['value'] / 1e6, d.get('verified')))" >> $OUT
  done
done
cat $OUT
