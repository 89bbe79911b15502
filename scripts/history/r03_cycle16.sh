#!/bin/bash
# round 3, cycle 16: after the removal of the slot-per-wave kernel: GPU suite; cfg1 per-kernel times alone
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $O/r03c16_gpu_tests.log 2>&1 || { tail -60 $O/r03c16_gpu_tests.log; exit 1; }
tail -2 $O/r03c16_gpu_tests.log
export WARPSTR_BENCH_PROFILING=1
cd /tmp && export TMPDIR=/tmp
WSX_STREAMS=1 WSX_CHUNKS=1 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/r03c16_alone -o p -- python3 $R/bench.py --workload cfg1 --no-cpu-baseline --no-verify --steps 6 --warmup 2 > $O/r03c16_alone.log 2>&1 || { tail $O/r03c16_alone.log; exit 1; }
python3 $R/scripts/kstats.py $O/r03c16_alone/p_kernel_trace.csv 9
WSX_STREAMS=1 WSX_CHUNKS=1 timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/r03c16_pmc -o p -- python3 $R/bench.py --workload cfg1 --no-cpu-baseline --no-verify --steps 2 --warmup 1 > $O/r03c16_pmc.log 2>&1 || { tail -5 $O/r03c16_pmc.log; exit 1; }
python3 - <<PY
import csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open('$O/r03c16_pmc/p_counter_collection.csv')):
    k = r['Kernel_Name'].split('(')[0].replace('void ','').replace('(anonymous namespace)::','')
    if k.startswith('at::') or k.startswith('__amd'): continue
    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print(f'{k:48s}', {c: f'{sum(x)/len(x):.3e}' for c, x in v.items() if c in ('SQ_INSTS_VALU','SQ_INSTS_SALU','SQ_WAVE_CYCLES')}, len(v['SQ_INSTS_VALU']))
PY
