#!/bin/bash
# round 4, cycle 10: the generated fill for one pass only (WSX_TUNE_GENERATED_PASSES) in the pipelined step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export WARPSTR_CACHE_DIR=$O/fillgen_cache
for rep in 1 2; do for f in "" "--generated-fill --generated-passes 1" "--generated-fill --generated-passes 3" "--generated-fill --generated-passes 2"; do
  WARPSTR_BENCH_PROFILING=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary $f 2>$O/r04c10_bench.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read())
print('[$f] %.3f ms/step  %.4g reads/s  fill alone %.3f ms  verified %s' % (d['ms_per_step'], d['value'], d['valu_roofline']['launch_ms_alone'], d['verified']['mismatches']))" || tail -5 $O/r04c10_bench.err
done; done | tee $O/r04c10_ab.log
