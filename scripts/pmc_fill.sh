#!/bin/bash
# PMC counters of the fill kernel on the headline shape (one pass per counter group).  Usage: scripts/pmc_fill.sh TAG [lib]
TAG=$1; R=$GRAFT_REPO_ROOT
[ -n "$2" ] && export WARPSTR_HIP_LIB=$R/build/exp/lib$2.so
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC" \
           "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_IFETCH"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/${TAG}_pmc$i -o p -- python3 $R/scripts/exp_shapes.py cfg3 50000 > $R/gpurun_out/${TAG}_pmc$i.log 2>&1 || { tail -5 $R/gpurun_out/${TAG}_pmc$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob('$R/gpurun_out/${TAG}_pmc*/p_counter_collection.csv')):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'dtw_fill' in r['Kernel_Name']:
            acc[r['Counter_Name']].append((float(r['Counter_Value']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
    for k, v in acc.items():
        v = v[-2:]
        print(f'{k:26s}', ' '.join(f'{x:.4g} ({d/1e6:.2f} ms)' for x, d in v))
PY
