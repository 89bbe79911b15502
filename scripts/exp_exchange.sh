#!/bin/bash
# The exchange micro-benchmark with its counters.  Usage: scripts/exp_exchange.sh TAG   (build/exp/exp_exchange built beforehand)
TAG=$1; R=$GRAFT_REPO_ROOT
$R/build/exp/exp_exchange | tee $R/gpurun_out/${TAG}_exchange.log
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  d=$R/gpurun_out/${TAG}_exchange_pmc_$(echo $set | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $d -o p -- $R/build/exp/exp_exchange > /dev/null 2>&1 || exit 1
  python3 - <<PY | tee -a $R/gpurun_out/${TAG}_exchange.log
import csv, collections, glob
rows=list(csv.DictReader(open(glob.glob('$d/**/p_counter_collection.csv', recursive=True)[0])))
by=collections.OrderedDict()
for r in rows:
    by.setdefault(int(r['Dispatch_Id']),{})[r['Counter_Name']]=float(r['Counter_Value']); by[int(r['Dispatch_Id'])]['k']=r['Kernel_Name']
ds=[by[k] for k in sorted(by) if 'rowloop' in by[k]['k']]
wave_rows = 512 * 4 * 4000.0
for d in ds[1::4]:
    print(d['k'][:24], ' '.join(f"{k}/wave-row={v / wave_rows:.2f}" for k, v in d.items() if k != 'k'))
PY
done
