#!/bin/bash
# round 3, cycle 10: run_raw on the device, multi-rank main_wrapper, bench secondary legs -- GPU suite, then the default bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $O/r03c10_gpu_tests.log 2>&1 || { tail -80 $O/r03c10_gpu_tests.log; exit 1; }
tail -2 $O/r03c10_gpu_tests.log
/usr/bin/time -v timeout -k 10 600 python bench.py > $O/r03c10_bench.json 2> $O/r03c10_bench.err || { tail -20 $O/r03c10_bench.err; exit 1; }
grep -E "Elapsed|Maximum resident" $O/r03c10_bench.err
python3 - <<PY
import json
d=json.load(open('$O/r03c10_bench.json'))
print('headline', round(d['value']), round(d['ms_per_step'],3), d['verified']['mismatches'], d['cpu_baseline'])
for k,v in d['secondary'].items():
    print(k, {x: (round(y,3) if isinstance(y,float) else y) for x,y in v.items() if x in ('value','ms_per_step','kernels','called_ok','ms_per_step_hbm_int16','ms_per_step_host_int16','reads_per_s_hbm_int16','reads_per_s_host_int16')}, v.get('verified',{}).get('mismatches'), v.get('identical_to_f64_path'))
PY
