#!/bin/bash
R=$GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $R/gpurun_out/r02c1_gpu_tests.log 2>&1 || { tail -40 $R/gpurun_out/r02c1_gpu_tests.log; exit 1; }
tail -2 $R/gpurun_out/r02c1_gpu_tests.log
for w in cfg1 cfg5; do
  timeout -k 10 300 python bench.py --workload $w > $R/gpurun_out/r02c1_bench_$w.json 2> $R/gpurun_out/r02c1_bench_$w.err || { tail $R/gpurun_out/r02c1_bench_$w.err; exit 1; }
  python3 -c "import json; d=json.load(open('$R/gpurun_out/r02c1_bench_$w.json')); print('$w', d['value'], d['ms_per_step'], d['roofline']['kernels'], d['valu_roofline'], d.get('verified'), d.get('cpu_baseline'))"
done
scripts/profile_round.sh r02c1_cfg1 --workload cfg1 && scripts/profile_round.sh r02c1_cfg5 --workload cfg5
