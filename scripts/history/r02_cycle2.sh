#!/bin/bash
R=$GRAFT_REPO_ROOT; TAG=$1
export WARPSTR_BENCH_PROFILING=1   # the kernels changed: profiles/fill_pmc.json is regenerated from this run
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $R/gpurun_out/${TAG}_gpu_tests.log 2>&1 || { tail -40 $R/gpurun_out/${TAG}_gpu_tests.log; exit 1; }
tail -2 $R/gpurun_out/${TAG}_gpu_tests.log
timeout -k 10 600 python scripts/fuzz_parity.py 6 > $R/gpurun_out/${TAG}_fuzz.log 2>&1 || { tail -20 $R/gpurun_out/${TAG}_fuzz.log; exit 1; }
tail -3 $R/gpurun_out/${TAG}_fuzz.log
export WARPSTR_BENCH_PROFILING=1
for w in headline cfg1 cfg5; do
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_$w.json 2> $R/gpurun_out/${TAG}_bench_$w.err || { tail $R/gpurun_out/${TAG}_bench_$w.err; exit 1; }
  python3 -c "import json; d=json.load(open('$R/gpurun_out/${TAG}_bench_$w.json')); print('$w', d['value'], d['ms_per_step'], d['roofline']['kernels'], d['valu_roofline']['launch_ms_alone'], d.get('verified'))"
done
for w in headline cfg1 cfg5; do scripts/profile_round.sh ${TAG}_$w --workload $w > $R/gpurun_out/${TAG}_profile_$w.log 2>&1 || { tail $R/gpurun_out/${TAG}_profile_$w.log; exit 1; }; done
echo profiles done
