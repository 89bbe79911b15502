#!/bin/bash
# round 4, final build: the random-locus sweeps (default settings, caller settings cycling, threshold > 1) and the random-read sweep
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 500 python scripts/fuzz_loci.py 1500 48 > $O/r04_final_fuzz_loci_1500.log 2>&1 || { tail -20 $O/r04_final_fuzz_loci_1500.log; exit 1; }
tail -1 $O/r04_final_fuzz_loci_1500.log
timeout -k 10 400 python scripts/fuzz_loci.py 600 32 --configs > $O/r04_final_fuzz_loci_configs_600.log 2>&1 || { tail -20 $O/r04_final_fuzz_loci_configs_600.log; exit 1; }
tail -1 $O/r04_final_fuzz_loci_configs_600.log
timeout -k 10 400 python scripts/fuzz_loci.py 400 32 --smooth > $O/r04_final_fuzz_loci_smooth_400.log 2>&1 || { tail -20 $O/r04_final_fuzz_loci_smooth_400.log; exit 1; }
tail -1 $O/r04_final_fuzz_loci_smooth_400.log
timeout -k 10 500 python scripts/fuzz_parity.py 10 > $O/r04_final_fuzz_parity_315k_reads.log 2>&1 || { tail -20 $O/r04_final_fuzz_parity_315k_reads.log; exit 1; }
tail -1 $O/r04_final_fuzz_parity_315k_reads.log
