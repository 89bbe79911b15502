#!/bin/bash
# PMC counters of the fill kernels on the example loci at flank 110 (scripts/exp_real_loci.py; one pass per counter group).  Usage: scripts/pmc_real_loci.sh TAG
TAG=$1; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  WSX_STREAMS=1 timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/${TAG}_pmc$i -o p -- python3 $R/scripts/exp_real_loci.py 20000 3000 > $R/gpurun_out/${TAG}_pmc$i.log 2>&1 || { tail -5 $R/gpurun_out/${TAG}_pmc$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('$R/gpurun_out/${TAG}_pmc*/p_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if 'dtw_fill' in r['Kernel_Name']:
            nm = r['Kernel_Name'].split('dtw_fill_fast')[1].split('(')[0]
            acc[nm][r['Counter_Name']].append((float(r['Counter_Value']), int(r['End_Timestamp']) - int(r['Start_Timestamp']), int(r.get('Grid_Size', r.get('Grid_Size_X', 0)) or 0)))
out = []
for nm, cs in acc.items():
    g = {k: (sum(x for x, _, _ in v) / len(v), sum(d for _, d, _ in v) / len(v) / 1e6) for k, v in cs.items()}
    if 'SQ_WAVES' not in g: continue
    w = g['SQ_WAVES'][0]; rows = w * 3000
    line = (f"{nm:28s} waves/launch {w:8.0f}  launch {g['SQ_INSTS_VALU'][1]:6.2f} ms  VALU/row {g['SQ_INSTS_VALU'][0] / rows:6.2f}  LDS inst/row {g['SQ_INSTS_LDS'][0] / rows:5.2f}  "
            f"SALU/row {g['SQ_INSTS_SALU'][0] / rows:5.2f}  VALU busy {g['SQ_ACTIVE_INST_VALU'][0] * 4 / g['SQ_BUSY_CYCLES'][0] / 4 if False else g['SQ_ACTIVE_INST_VALU'][0] / (g['SQ_WAVE_CYCLES'][0] if False else 1):.3g}")
    out.append(line)
    print(nm, {k: (round(v[0]), round(v[1], 2)) for k, v in g.items()})
PY
