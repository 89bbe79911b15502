#!/bin/bash
# round 3: randomised parity sweeps on the round's final kernels (logs kept under profiles/)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout -k 10 900 python scripts/fuzz_loci.py 4000 48 > $O/r03_fuzz_loci_4000.log 2>&1 || { tail -30 $O/r03_fuzz_loci_4000.log; exit 1; }
tail -1 $O/r03_fuzz_loci_4000.log
timeout -k 10 900 python scripts/fuzz_parity.py 30 > $O/r03_fuzz_parity_945k_reads.log 2>&1 || { tail -30 $O/r03_fuzz_parity_945k_reads.log; exit 1; }
tail -1 $O/r03_fuzz_parity_945k_reads.log
timeout -k 10 900 python scripts/fuzz_loci.py 1500 48 --configs > $O/r03_fuzz_loci_configs_1500.log 2>&1 || { tail -30 $O/r03_fuzz_loci_configs_1500.log; exit 1; }
tail -1 $O/r03_fuzz_loci_configs_1500.log
