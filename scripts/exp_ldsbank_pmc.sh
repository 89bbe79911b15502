#!/bin/bash
# LDS bank conflicts of exp_ldsbank's tables from the hardware counter.  Usage: scripts/exp_ldsbank_pmc.sh TAG
TAG=$1; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/${TAG}_ldsbank -o p -- $R/build/exp/exp_ldsbank > $R/gpurun_out/${TAG}_ldsbank.log 2>&1 || { tail $R/gpurun_out/${TAG}_ldsbank.log; exit 1; }
python3 - <<PY
import csv, collections
R='$R'
rows=list(csv.DictReader(open(R+'/gpurun_out/${TAG}_ldsbank/p_counter_collection.csv')))
by=collections.OrderedDict()
for r in rows:
    by.setdefault(int(r['Dispatch_Id']),{})[r['Counter_Name']]=float(r['Counter_Value']); by[int(r['Dispatch_Id'])]['k']=r['Kernel_Name']
names=[l.split('  ')[0] for l in open(R+'/gpurun_out/${TAG}_ldsbank.log') if 'ns per wave' in l]
ds=[by[k] for k in sorted(by) if by[k]['k'].startswith('void k<')]
out=[]
for i in range(0,len(ds),4):
    w=ds[i+1]; r=ds[i+3]; nm=names[i//4]
    line=f"{nm[:96]:96s} write: conflict {w['SQ_LDS_BANK_CONFLICT']/w['SQ_INSTS_LDS']:.2f} active {w['SQ_ACTIVE_INST_LDS']/w['SQ_INSTS_LDS']:.2f} | read: conflict {r['SQ_LDS_BANK_CONFLICT']/r['SQ_INSTS_LDS']:.2f} active {r['SQ_ACTIVE_INST_LDS']/r['SQ_INSTS_LDS']:.2f}"
    print(line); out.append(line)
open(R+'/gpurun_out/${TAG}_lds_bank_rule.log','w').write('scripts/exp_ldsbank.hip under rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS (counter units per ds_write_b64 / ds_read_b64 of a wavefront)\n'+'\n'.join(out)+'\n')
PY
