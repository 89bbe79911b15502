#!/bin/bash
# round 4, cycle 9: the default bench run as the driver launches it (all legs), timed; new 4-rank rehearsal test
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_bench_contract.py -x -q -k "four_ranks" > $O/r04c9_tests.log 2>&1 || { tail -30 $O/r04c9_tests.log; exit 1; }
tail -1 $O/r04c9_tests.log
T0=$(date +%s.%N); timeout -k 10 900 python bench.py > $O/r04c9_bench.json 2> $O/r04c9_bench.err || { tail -20 $O/r04c9_bench.err; exit 1; }
echo "bench wall $(echo "$(date +%s.%N) - $T0" | bc) s"
python - <<PY
import json
d=json.load(open('$O/r04c9_bench.json'))
print('reads/s', d['value'], 'ms/step', d['ms_per_step'], 'valu', d['valu_roofline'].get('frac'), d['valu_roofline'].get('counters'))
print({k:(v['ms_per_step'],v['value']) for k,v in d['secondary'].items() if 'ms_per_step' in v})
print('generated', json.dumps(d['secondary']['generated_fill']))
m=d['many_loci']; print('many_loci', m['loci_per_s'], m['ms_per_locus'], json.dumps(m['one_handle']), m['one_handle_per_locus']['ms_per_locus'], m['outputs_identical'])
print('cfg5 driver', json.dumps(d['secondary']['cfg5']['through_driver']))
PY
