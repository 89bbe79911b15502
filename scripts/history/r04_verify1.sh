#!/bin/bash
# round 4: the whole GPU suite on the current build, then the random-locus and random-read parity sweeps
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export WARPSTR_CACHE_DIR=$O/fillgen_cache
WARPSTR_BENCH_PROFILING=1 timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r04_gpu_tests.log 2>&1 || { tail -40 $O/r04_gpu_tests.log; exit 1; }
tail -1 $O/r04_gpu_tests.log
timeout -k 10 400 python scripts/fuzz_loci.py 1000 48 > $O/r04_fuzz_loci_1000.log 2>&1 || { tail -20 $O/r04_fuzz_loci_1000.log; exit 1; }
tail -3 $O/r04_fuzz_loci_1000.log
timeout -k 10 400 python scripts/fuzz_parity.py 6 > $O/r04_fuzz_parity_189k_reads.log 2>&1 || { tail -20 $O/r04_fuzz_parity_189k_reads.log; exit 1; }
tail -3 $O/r04_fuzz_parity_189k_reads.log
