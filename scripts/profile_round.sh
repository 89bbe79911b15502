#!/bin/bash
# Profiles of the bench command for profiles/: kernel trace + stats (same command as the bench line), then separate PMC passes
# (FETCH_SIZE, WRITE_SIZE, SQ counters) as MI355X_MICROARCH.md prescribes.  Usage: scripts/profile_round.sh TAG [bench.py arguments, e.g. --workload cfg1]
TAG=$1; shift; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export WARPSTR_BENCH_PROFILING=1  # bench.py: do not insist on an existing PMC entry for this kernel
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o p -- python3 $R/bench.py --no-cpu-baseline "$@" > $O/bench_under_trace.json 2> $O/trace.err || { tail -5 $O/trace.err; exit 1; }
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
  "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
  "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT" ; do
  i=$((i+1))
  WSX_STREAMS=1 timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $O/pmc$i -o p -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > $O/pmc$i.json 2> $O/pmc$i.err || { tail -5 $O/pmc$i.err; exit 1; }
done
ls $O
