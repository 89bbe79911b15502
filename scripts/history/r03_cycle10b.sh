#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
start=$(date +%s.%N)
timeout -k 10 600 python bench.py > $O/r03c10_bench.json 2> $O/r03c10_bench.err || { tail -20 $O/r03c10_bench.err; exit 1; }
end=$(date +%s.%N); echo "bench.py wall: $(python3 -c "print(round($end-$start,1))") s"
python3 - <<PY
import json
d=json.load(open('$O/r03c10_bench.json'))
print('headline', round(d['value']), round(d['ms_per_step'],3), d['verified']['mismatches'], d['cpu_baseline'])
for k,v in d['secondary'].items():
    print(k, {x: (round(y,3) if isinstance(y,float) else y) for x,y in v.items() if x in ('value','ms_per_step','kernels','called_ok','ms_per_step_hbm_int16','ms_per_step_host_int16','reads_per_s_hbm_int16','reads_per_s_host_int16')}, v.get('verified',{}).get('mismatches'), v.get('identical_to_f64_path'))
PY
