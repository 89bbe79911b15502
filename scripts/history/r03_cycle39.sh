#!/bin/bash
# round 3, cycle 39: short_read_kernel's normalising division with one refined reciprocal per read: loader parity, from-raw leg
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "loader or raw or wrapper or upstream" > $O/r03c39_tests.log 2>&1; rc=$?; tail -3 $O/r03c39_tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python scripts/fuzz_loader.py 200 > $O/r03c39_fuzz_loader.log 2>&1; rc=$?; tail -2 $O/r03c39_fuzz_loader.log; [ $rc -eq 0 ] || exit $rc
export WARPSTR_BENCH_PROFILING=1
for rep in 1 2 3; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --from-raw --steps 10 > $O/r03c39_b.json 2> $O/r03c39_b.err || { tail $O/r03c39_b.err; exit 1; }
  python3 -c "import json; d=json.load(open('$O/r03c39_b.json')); f=d['from_raw']; print(round(d['ms_per_step'],3), 'from raw (HBM int16):', round(f['ms_per_step_hbm_int16'],3), round(f['reads_per_s_hbm_int16']), 'host int16:', round(f['ms_per_step_host_int16'],3), f['identical_to_f64_path'])"
done
