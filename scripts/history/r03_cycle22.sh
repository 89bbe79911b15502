#!/bin/bash
# round 3, cycle 22: bench contract tests after the roofline bookkeeping change, then the bench lines again
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bench_contract.py -m gpu -q -x > $O/r03c22_gpu_tests.log 2>&1 || { tail -60 $O/r03c22_gpu_tests.log; exit 1; }
tail -2 $O/r03c22_gpu_tests.log
scripts/r03_lines.sh
timeout -k 10 700 python scripts/fuzz_loci.py 1500 48 --configs > $O/r03_fuzz_loci_configs_1500.log 2>&1 || { tail -30 $O/r03_fuzz_loci_configs_1500.log; exit 1; }
tail -1 $O/r03_fuzz_loci_configs_1500.log
