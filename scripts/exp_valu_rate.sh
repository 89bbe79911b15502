#!/bin/bash
# Cycles per fp64 VALU instruction from the hardware counters.  Usage: scripts/exp_valu_rate.sh TAG
TAG=$1; R=$GRAFT_REPO_ROOT
$R/build/exp/exp_valu_rate | tee $R/gpurun_out/${TAG}_valu_rate.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/${TAG}_valu_rate -o p -- $R/build/exp/exp_valu_rate > /dev/null 2>&1 || exit 1
python3 - <<PY | tee -a $R/gpurun_out/${TAG}_valu_rate.log
import csv, collections
rows=list(csv.DictReader(open('$R/gpurun_out/${TAG}_valu_rate/p_counter_collection.csv')))
by=collections.OrderedDict()
for r in rows:
    by.setdefault(int(r['Dispatch_Id']),{})[r['Counter_Name']]=float(r['Counter_Value']); by[int(r['Dispatch_Id'])]['k']=r['Kernel_Name']
ds=[by[k] for k in sorted(by) if 'k<' in by[k]['k']]
for d in ds[1::2]:
    cyc=d['GRBM_GUI_ACTIVE']/8.0
    print(f"{d['k'][:40]:40s} SIMD cycles per VALU wave-instruction: {cyc*1024/d['SQ_INSTS_VALU']:.2f}   (SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = {d['SQ_ACTIVE_INST_VALU']/d['SQ_INSTS_VALU']:.2f})")
PY
