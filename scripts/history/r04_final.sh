#!/bin/bash
# round 4, final: the GPU suite as the driver runs it, then the default bench line (all legs) with its wall-clock
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r04_final_gpu_tests.log 2>&1 || { tail -40 $O/r04_final_gpu_tests.log; exit 1; }
tail -1 $O/r04_final_gpu_tests.log
T0=$(date +%s)
timeout -k 10 900 python bench.py > $O/r04_final_bench.json 2> $O/r04_final_bench.err || { tail -20 $O/r04_final_bench.err; exit 1; }
echo "bench wall $(( $(date +%s) - T0 )) s"
python - <<PY
import json
d=json.load(open('$O/r04_final_bench.json'))
print('reads/s', d['value'], 'ms/step', d['ms_per_step'], 'roofline', d['roofline']['frac'], 'valu', d['valu_roofline'].get('frac'), d['valu_roofline'].get('counters_from'))
print({k:(v['ms_per_step'],v['value'],v.get('traffic_over_algorithmic')) for k,v in d['secondary'].items() if 'ms_per_step' in v})
m=d['many_loci']; print('many_loci', m['loci_per_s'], m['ms_per_locus'], json.dumps(m['one_handle']['per_locus_ms']), m['one_handle']['once_s'], m['one_handle_per_locus']['ms_per_locus'], m['outputs_identical'])
print('cfg5 driver', json.dumps(d['secondary']['cfg5']['through_driver']))
PY
