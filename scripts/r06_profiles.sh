#!/bin/bash
# Round 6's profile cycle: the bench command under kernel trace + the PMC passes (the caller's kernels are round 4's: the counters
# are re-collected all the same, on the round's last library), then the new kernels -- wsx_zstd_decode beside wsx_vbz_decode in a
# from_fast5 run in one process, and their counters per launch.   Usage: scripts/r06_profiles.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
bash $R/scripts/profile_round.sh r06_headline > $O/r06_headline_profile.log 2>&1 || { tail -5 $O/r06_headline_profile.log; exit 1; }
(cd $R && python scripts/summarize_profiles.py $O/r06_headline r06_headline) > $O/r06_headline_summary.log 2>&1 || { tail -5 $O/r06_headline_summary.log; exit 1; }
cd /tmp && export TMPDIR=/tmp
export WARPSTR_BENCH_FAST5_ONLY=one_process
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fast5 -o fast5 -- python3 $R/scripts/exp_from_fast5.py 600 > $O/r06_from_fast5_under_trace.json 2> $O/r06_from_fast5_under_trace.err
find $O/prof_fast5 -name '*kernel_stats.csv' -exec cp {} $O/r06_from_fast5_kernel_stats.csv \;
unset WARPSTR_BENCH_FAST5_ONLY
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $O/prof_zstd_$tag -o z -- python3 $R/scripts/prof_zstd.py 2048 4 > /dev/null 2>&1 || exit 1
done
python3 - <<PY | tee $O/r06_zstd_pmc.log
import csv, glob, collections
O='$O'
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for tag in ('FETCH_SIZE','WRITE_SIZE','GRBM_GUI_ACTIVE'):
    for f in glob.glob(f'{O}/prof_zstd_{tag}/**/z_counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0].split('::')[-1]
            if k.startswith('zstd_'): agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
print('per launch of 2 048 real chunks (mean over the launches of scripts/prof_zstd.py 2048 4; separate --pmc passes;')
print('FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them, FETCH_SIZE doubled for 16-byte loads as MI355X_MICROARCH.md prescribes):')
for k,c in agg.items():
    line=[k]
    for name,vals in c.items():
        v=sum(vals)/len(vals)
        line.append(f'{name}={v:.4g}')
    print('  '+'  '.join(line))
PY
ls $O | grep r06_
