#!/bin/bash
# round 3: the bench lines kept under profiles/ (the default run with its secondary legs; cfg1; cfg5), counters quoted from
# profiles/fill_pmc.json
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for w in headline cfg1 cfg5; do
  extra="--workload $w"; [ $w = headline ] && extra=""
  timeout -k 10 600 python bench.py $extra > $O/r03_bench_$w.json 2> $O/r03_bench_$w.err || { tail $O/r03_bench_$w.err; exit 1; }
  python3 -c "import json; d=json.load(open('$O/r03_bench_$w.json')); print('$w', round(d['value']), round(d['ms_per_step'],3), d['roofline']['kernels'], d['valu_roofline'].get('frac_at_observed_clock'), d.get('verified',{}).get('mismatches'), d.get('cpu_baseline',{}).get('value'), {k: round(v.get('ms_per_step', v.get('ms_per_step_hbm_int16', 0)), 3) for k, v in d.get('secondary', {}).items()})"
done
