#!/bin/bash
# round 4, cycle 7: generated fill, second iteration (register pressure, LDS-staged traceback): tests, A/B, single-stream kernel times
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export WARPSTR_CACHE_DIR=$O/fillgen_cache
timeout -k 10 600 python -m pytest tests/test_gpu_generated_fill.py -x -q > $O/r04c7_gen_tests.log 2>&1 || { tail -60 $O/r04c7_gen_tests.log; exit 1; }
tail -1 $O/r04c7_gen_tests.log
for rep in 1 2; do for f in "" "--builtin-fill"; do
  WARPSTR_BENCH_PROFILING=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary $f 2>$O/r04c7_bench.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read())
print('[$f] %.3f ms/step  %.4g reads/s  fill alone %.3f ms  verified %s  fill %s' % (d['ms_per_step'], d['value'], d['valu_roofline']['launch_ms_alone'], d['verified']['mismatches'], d['config']['fill']['kind'][:9]))" || tail -5 $O/r04c7_bench.err
done; done | tee $O/r04c7_ab.log
cd /tmp && export TMPDIR=/tmp
WARPSTR_BENCH_PROFILING=1 WSX_STREAMS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04c7_prof -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/r04c7_prof.log 2>&1
python3 $R/scripts/kstats.py $O/r04c7_prof/p_kernel_trace.csv 5
