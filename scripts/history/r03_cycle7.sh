#!/bin/bash
# round 3, cycle 7: segmentation kernels alone (one stream, one chunk): durations and instruction counts
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
export WARPSTR_BENCH_PROFILING=1
cd /tmp && export TMPDIR=/tmp
for env in "WSX_SEGMENT_BLOCK_KERNEL=1" "WSX_X=0"; do
  tag=$(echo $env | tr -d ' =' )
  export $env
  WSX_STREAMS=1 WSX_CHUNKS=1 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/r03c7_alone_$tag -o p -- python3 $R/bench.py --no-cpu-baseline --no-verify --steps 6 --warmup 2 > $O/r03c7_alone_$tag.log 2>&1 || { tail $O/r03c7_alone_$tag.log; exit 1; }
  echo "== $env (ms per step, 9 steps)"; python3 $R/scripts/kstats.py $O/r03c7_alone_$tag/p_kernel_trace.csv 9
  WSX_STREAMS=1 WSX_CHUNKS=1 timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU --output-format csv -d $O/r03c7_pmc_$tag -o p -- python3 $R/bench.py --no-cpu-baseline --no-verify --steps 2 --warmup 1 > $O/r03c7_pmc_$tag.log 2>&1 || { tail -5 $O/r03c7_pmc_$tag.log; exit 1; }
  python3 - <<PY
import csv, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open('$O/r03c7_pmc_$tag/p_counter_collection.csv')):
    if 'segment' in r['Kernel_Name']:
        acc[r['Counter_Name']].append((float(r['Counter_Value']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
for k, v in acc.items():
    v = v[-2:]
    print(f'{k:26s}', ' '.join(f'{x:.5g} ({d/1e6:.2f} ms)' for x, d in v))
PY
  unset WSX_SEGMENT_BLOCK_KERNEL
done
