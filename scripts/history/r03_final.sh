#!/bin/bash
# round 3: profiles of the three bench workloads (kernel trace + PMC passes) for the round's final library; afterwards
# scripts/summarize_profiles.py gpurun_out/r03f_<w> r03_<w> for each, then scripts/r03_lines.sh for the bench lines
scripts/profile_round.sh r03f_headline && scripts/profile_round.sh r03f_cfg1 --workload cfg1 && scripts/profile_round.sh r03f_cfg5 --workload cfg5
