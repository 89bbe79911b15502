#!/bin/bash
# A/B of environment settings on the bench line, interleaved repeats.  Usage: scripts/ab_env.sh TAG REPEATS "ENV1" "ENV2" ... [-- bench args]
TAG=$1; REP=$2; shift 2
R=$GRAFT_REPO_ROOT; L=$R/gpurun_out/${TAG}_ab.log; : > $L
ENVS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do ENVS+=("$1"); shift; done; [ "$1" = "--" ] && shift
export WARPSTR_BENCH_PROFILING=1
for r in $(seq $REP); do for e in "${ENVS[@]}"; do
  out=$(env $e timeout -k 10 200 python $R/bench.py --no-cpu-baseline --no-verify "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step  %.4g reads/s  alone %.3f' % (d['ms_per_step'], d['value'], d['valu_roofline']['launch_ms_alone']))") || exit 1
  echo "[$e] $out" | tee -a $L
done; done
