#!/bin/bash
# SQ counters of vbz_decode_kernel (scripts/prof_vbz.py: 2 048 real blocks per launch): vector instructions and LDS instructions per
# launch, busy cycles -- is the kernel waiting for memory or issuing?
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS"; do
  tag=$(echo $grp | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_vbz_$tag -o p -- python3 $R/scripts/prof_vbz.py 2048 4 > $O/pmc_vbz.json 2> $O/pmc_vbz.err || { tail -5 $O/pmc_vbz.err; exit 1; }
done
python3 - <<'PY'
import csv, glob, os, collections
O=os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out'
acc=collections.defaultdict(list)
for f in glob.glob(O+'/pmc_vbz_*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'vbz_decode' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
with open(O+'/r05_vbz_pmc.log','w') as out:
    for k,v in sorted(acc.items()):
        line=f'{k:28s} per launch {sum(v)/len(v):.4g}  ({len(v)} launches)'
        print(line); out.write(line+'\n')
PY
