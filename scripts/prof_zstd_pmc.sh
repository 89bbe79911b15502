#!/bin/bash
# Counters of the zstd kernels per launch of 2 048 real chunks (scripts/prof_zstd.py 2048 4), separate --pmc passes, and the kernel
# trace of the same command.  Usage (GPU box): scripts/prof_zstd_pmc.sh   -> gpurun_out/zstd_pmc/summary.log
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/zstd_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o z -- python3 $R/scripts/prof_zstd.py 2048 4 > $O/prof_zstd.json 2> /dev/null || exit 1
find $O/trace -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats.csv \;
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$tag -o z -- python3 $R/scripts/prof_zstd.py 2048 4 > /dev/null 2>&1 || exit 1
done
python3 - <<PY | tee $O/summary.log
import csv, glob, collections, re
O='$O'
print('scripts/prof_zstd_pmc.sh: per launch of 2 048 real chunks of the upstream file (scripts/prof_zstd.py 2048 4 under rocprofv3, separate --pmc passes,')
print('mean over the launches; FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them -- FETCH_SIZE under-counts 16-byte loads by half on gfx950,')
print('MI355X_MICROARCH.md; GRBM_GUI_ACTIVE and the SQ counters summed over the eight XCDs):')
for r in csv.DictReader(open(f'{O}/kernel_stats.csv')):
    if 'zstd_' in r['Name']:
        print(f"  {re.search(r'(zstd_[a-z]+_kernel)', r['Name']).group(1):28s} {int(r['Calls']):3d} launches, {float(r['AverageNs'])/1e6:.3f} ms each")
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f'{O}/pmc_*/**/z_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m=re.search(r'(zstd_[a-z]+_kernel)', r['Kernel_Name'])
        if m: agg[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg):
    print('  '+k+': '+'  '.join(f'{n}={sum(v)/len(v):.4g}' for n,v in sorted(agg[k].items())))
PY
