#!/bin/bash
# A/B of experiment libraries on the headline bench (no CPU baseline).  Usage: ab_test.sh "lib1 lib2 ..." ["ENV=val ..."]
R=$GRAFT_REPO_ROOT
for v in $1; do
  for rep in 1 2; do
    out=$(env WARPSTR_HIP_LIB=$R/build/exp/lib$v.so $2 timeout -k 10 120 python $R/bench.py --no-cpu-baseline --steps 5 --warmup 1 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step  %.3g reads/s  fill alone %.3f ms ok=%d' % (d['ms_per_step'], d['value'], d['valu_roofline']['launch_ms_alone'], d['config']['called_ok']))")
    echo "$v $2: $out"
  done
done
