#!/bin/bash
# round 5: the profiles of the round (kernel trace + stats of the bench command, PMC passes) for the three workloads, then the
# counters of the example loci's fills (the slot-major placement's walk changed: fill_pmc.json is re-measured on these sources)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
scripts/profile_round.sh r05_headline || exit 1
scripts/profile_round.sh r05_cfg1 --workload cfg1 || exit 1
scripts/profile_round.sh r05_cfg5 --workload cfg5 || exit 1
scripts/pmc_real_loci.sh r05_real > $O/r05_real_loci_pmc.log 2>&1 || { tail -5 $O/r05_real_loci_pmc.log; exit 1; }
tail -6 $O/r05_real_loci_pmc.log
# keep what travels back small: the traces' per-dispatch tables of the PMC passes are what summarize_profiles.py reads
find $O/r05_* -name '*.db' -delete 2>/dev/null
du -sh $O/r05_headline $O/r05_cfg1 $O/r05_cfg5 2>/dev/null
