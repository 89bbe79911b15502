#!/bin/bash
# round 3, cycle 26: fill row loop without the 64-row chopping -- parity, random loci, then headline / cfg1 / cfg5
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
export WARPSTR_BENCH_PROFILING=1
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipelined.py -m gpu -q -x > $O/r03c26_gpu_tests.log 2>&1 || { tail -60 $O/r03c26_gpu_tests.log; exit 1; }
tail -1 $O/r03c26_gpu_tests.log
timeout -k 10 600 python scripts/fuzz_loci.py 800 48 > $O/r03c26_fuzz_loci.log 2>&1 || { tail -30 $O/r03c26_fuzz_loci.log; exit 1; }
tail -1 $O/r03c26_fuzz_loci.log
timeout -k 10 600 python scripts/fuzz_parity.py 6 > $O/r03c26_fuzz_parity.log 2>&1 || { tail -30 $O/r03c26_fuzz_parity.log; exit 1; }
tail -1 $O/r03c26_fuzz_parity.log
line() { python3 -c "import json,sys; d=json.load(open('$1')); print('$2', round(d['value']), round(d['ms_per_step'],3), 'fill union', round(d['roofline'].get('fill_union_ms_per_step',0),3), 'alone', round(d['valu_roofline']['launch_ms_alone'],3), d.get('verified',{}).get('mismatches'))"; }
for w in headline cfg1 cfg5 headline cfg1; do
  extra="--workload $w"; [ $w = headline ] && extra=""
  timeout -k 10 300 python bench.py $extra --no-cpu-baseline > $O/r03c26_b.json 2> $O/r03c26_b.err || { tail $O/r03c26_b.err; exit 1; }
  line $O/r03c26_b.json "$w"
done
