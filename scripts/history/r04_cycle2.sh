#!/bin/bash
# round 4, cycle 2: per-read back-pointer offsets (one region per chunk), threaded placement: GPU suite, many-loci timing, bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
WARPSTR_BENCH_PROFILING=1 timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/r04c2_gpu_tests.log 2>&1 || { tail -40 $O/r04c2_gpu_tests.log; exit 1; }
tail -1 $O/r04c2_gpu_tests.log
timeout -k 10 300 python scripts/exp_many_loci.py 2000 32 2>&1 | grep -v amdgpu.ids | tee $O/r04c2_many_loci.log
timeout -k 10 500 python bench.py > $O/r04c2_bench.json 2> $O/r04c2_bench.err || { tail -20 $O/r04c2_bench.err; exit 1; }
python - <<PY
import json
d=json.load(open('$O/r04c2_bench.json'))
print('reads/s', d['value'], 'ms/step', d['ms_per_step'], 'workspace', d['workspace'])
print({k:(v['ms_per_step'],v['value']) for k,v in d['secondary'].items() if 'ms_per_step' in v})
m=d['many_loci']; print('many_loci', m['loci_per_s'], m['ms_per_locus'], m['one_handle'], m['one_handle_per_locus'], m['outputs_identical'])
PY
