#!/bin/bash
# round 4, cycle 1: the several-loci driver on the GPU (new tests first), the whole GPU suite, the bench line with its new legs
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_loci.py -x -q > $O/r04c1_loci_tests.log 2>&1 || { tail -40 $O/r04c1_loci_tests.log; exit 1; }
tail -1 $O/r04c1_loci_tests.log
WARPSTR_BENCH_PROFILING=1 timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/r04c1_gpu_tests.log 2>&1 || { tail -40 $O/r04c1_gpu_tests.log; exit 1; }
tail -1 $O/r04c1_gpu_tests.log
timeout -k 10 500 python bench.py > $O/r04c1_bench.json 2> $O/r04c1_bench.err || { tail -20 $O/r04c1_bench.err; exit 1; }
python - <<PY
import json
d=json.load(open('$O/r04c1_bench.json'))
print('reads/s', d['value'], 'ms/step', d['ms_per_step'], 'workspace', d['workspace'])
print('many_loci', json.dumps(d.get('many_loci'), indent=1))
print('cfg5 driver', json.dumps(d['secondary']['cfg5'].get('through_driver'), indent=1))
PY
