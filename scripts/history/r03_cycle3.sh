#!/bin/bash
# round 3, cycle 3: the several-slot fills alone (one stream, one chunk): durations and PMC counters, cfg1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
export WARPSTR_BENCH_PROFILING=1
cd /tmp && export TMPDIR=/tmp
for env in "WSX_FILL_LM=0 WSX_FILL_WG=0" "WSX_FILL_LM=0 WSX_FILL_WG=1" "WSX_FILL_LM=1"; do
  tag=$(echo $env | tr -d ' =' )
  export $env
  WSX_STREAMS=1 WSX_CHUNKS=1 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/r03c3_alone_$tag -o p -- python3 $R/bench.py --workload cfg1 --no-cpu-baseline --no-verify --steps 6 --warmup 2 > $O/r03c3_alone_$tag.log 2>&1 || { tail $O/r03c3_alone_$tag.log; exit 1; }
  echo "== $env (ms per step, 9 steps)"; python3 $R/scripts/kstats.py $O/r03c3_alone_$tag/p_kernel_trace.csv 9
  i=0
  for grp in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
             "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT"; do
    i=$((i+1))
    WSX_STREAMS=1 WSX_CHUNKS=1 timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $O/r03c3_pmc${i}_$tag -o p -- python3 $R/bench.py --workload cfg1 --no-cpu-baseline --no-verify --steps 2 --warmup 1 > $O/r03c3_pmc${i}_$tag.log 2>&1 || { tail -5 $O/r03c3_pmc${i}_$tag.log; exit 1; }
  done
  python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob('$O/r03c3_pmc*_$tag/p_counter_collection.csv')):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'dtw_fill' in r['Kernel_Name']:
            acc[r['Counter_Name']].append((float(r['Counter_Value']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
    for k, v in acc.items():
        v = v[-2:]
        print(f'{k:26s}', ' '.join(f'{x:.5g} ({d/1e6:.2f} ms)' for x, d in v))
PY
done
