#!/bin/bash
# round 3, cycle 38: the general loader kernels look at 256 reads per block after short_read_kernel (from-raw leg A/B, loader parity)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "loader or raw or wrapper or upstream" > $O/r03c38_tests.log 2>&1; rc=$?; tail -3 $O/r03c38_tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python scripts/fuzz_loader.py 8 > $O/r03c38_fuzz_loader.log 2>&1; rc=$?; tail -3 $O/r03c38_fuzz_loader.log; [ $rc -eq 0 ] || exit $rc
export WARPSTR_BENCH_PROFILING=1
for rep in 1 2; do for env in "WSX_PREP_NO_GROUPS=1" "WSX_UNUSED=1"; do
  env $env timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --from-raw --steps 10 > $O/r03c38_b.json 2> $O/r03c38_b.err || { tail $O/r03c38_b.err; exit 1; }
  python3 -c "import json; d=json.load(open('$O/r03c38_b.json')); f=d['from_raw']; print('[$env]', round(d['ms_per_step'],3), 'from raw (HBM int16):', round(f['ms_per_step_hbm_int16'],3), round(f['reads_per_s_hbm_int16']), 'host int16:', round(f['ms_per_step_host_int16'],3), f['identical_to_f64_path'])"
done; done
