#!/bin/bash
# round 4: the profiles of the round (kernel trace + stats of the bench command, PMC passes) for the three workloads, then the
# counters of the example loci's fills
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
scripts/profile_round.sh r04_headline || exit 1
scripts/profile_round.sh r04_cfg1 --workload cfg1 || exit 1
scripts/profile_round.sh r04_cfg5 --workload cfg5 || exit 1
scripts/pmc_real_loci.sh r04_real > $O/r04_real_loci_pmc.log 2>&1 || { tail -5 $O/r04_real_loci_pmc.log; exit 1; }
tail -6 $O/r04_real_loci_pmc.log
