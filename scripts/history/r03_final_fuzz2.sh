#!/bin/bash
# round 3, final library: random-locus sweeps (default settings; caller settings cycling; rescaling.threshold > 1 with perturbed levels)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout -k 10 420 python scripts/fuzz_loci.py 1500 48 > $O/r03ff_loci.log 2>&1 || { tail -5 $O/r03ff_loci.log; exit 1; }
tail -1 $O/r03ff_loci.log
timeout -k 10 300 python scripts/fuzz_loci.py 800 32 --configs > $O/r03ff_configs.log 2>&1 || { tail -5 $O/r03ff_configs.log; exit 1; }
tail -1 $O/r03ff_configs.log
timeout -k 10 300 python scripts/fuzz_loci.py 1200 32 --smooth > $O/r03ff_smooth.log 2>&1 || { tail -5 $O/r03ff_smooth.log; exit 1; }
tail -1 $O/r03ff_smooth.log
