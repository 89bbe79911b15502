#!/bin/bash
# A/B of library builds on a bench workload, interleaved repeats.  Usage: scripts/ab_lib.sh TAG REPEATS "lib1.so lib2.so" [bench args]
TAG=$1; REP=$2; LIBS=$3; shift 3
R=$GRAFT_REPO_ROOT; L=$R/gpurun_out/${TAG}_ablib.log; : > $L
export WARPSTR_BENCH_PROFILING=1
for r in $(seq $REP); do for lib in $LIBS; do
  out=$(WARPSTR_HIP_LIB=$R/$lib timeout -k 10 200 python $R/bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step  %.4g reads/s  fill alone %.3f ms  verified %d/%d' % (d['ms_per_step'], d['value'], d['valu_roofline']['launch_ms_alone'], d['verified']['reads']-d['verified']['mismatches'], d['verified']['reads']))") || exit 1
  echo "[$lib $*] $out" | tee -a $L
done; done
