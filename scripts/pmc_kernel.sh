#!/bin/bash
# PMC counters of one kernel (substring match) for a command.  Usage: pmc_kernel.sh TAG KERNEL_SUBSTR -- cmd...
TAG=$1; KSUB=$2; shift 3
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/${TAG}_pmc$i -o p -- "$@" > $R/gpurun_out/${TAG}_pmc$i.log 2>&1 || { tail -5 $R/gpurun_out/${TAG}_pmc$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob('$R/gpurun_out/${TAG}_pmc*/p_counter_collection.csv')):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if '$KSUB' in r['Kernel_Name']:
            acc[r['Counter_Name']].append((float(r['Counter_Value']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
    for k, v in acc.items():
        v = v[-2:]
        print(f'{k:26s}', ' '.join(f'{x:.4g} ({d/1e6:.2f} ms)' for x, d in v))
PY
