"""Turn one scripts/profile_round.sh output directory into the files kept under profiles/.
Usage: summarize_profiles.py gpurun_out/<TAG> <name>   ->  profiles/<name>_kernel_stats.csv, <name>_pmc.json, <name>_traffic.json"""
import csv, json, os, sys, collections

src, name = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, 'profiles')

# per-kernel stats of the bench command itself (rocprofv3 --kernel-trace --stats); torch's own kernels dropped
rows = list(csv.DictReader(open(os.path.join(src, 'trace', 'p_kernel_stats.csv'))))
keep = [r for r in rows if 'at::native' not in r['Name'] and not r['Name'].startswith('__amd')]
with open(os.path.join(prof, f'{name}_kernel_stats.csv'), 'w', newline='') as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()), quoting=csv.QUOTE_NONNUMERIC)
    w.writeheader()
    w.writerows(keep)

# the dominant kernel's launches by size: bench.py times K = 20 steps (after 2 warm-up steps) of 4 chunks x 2 passes (25 000 reads per launch at the default
# size) and then runs one extra untimed single-stream step (2 launches of all reads) for the VALU roofline; rocprofv3's
# per-kernel average above mixes the two, bench.py's roofline.launch_ms is the first kind
tr = list(csv.DictReader(open(os.path.join(src, 'trace', 'p_kernel_trace.csv'))))
by = collections.defaultdict(list)
for r in tr:
    if 'dtw_fill' in r['Kernel_Name']:
        by[int(r['Grid_Size_X'])].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
with open(os.path.join(prof, f'{name}_fill_launches.txt'), 'w') as f:
    f.write('dtw_fill launches of the bench command by grid size (threads = 64 per read), from the kernel trace\n')
    for g, v in sorted(by.items()):
        f.write(f'grid {g:9d} threads = {g // 64:6d} reads per launch: {len(v):3d} launches, average {sum(v) / len(v) / 1e6:.4f} ms, '
                f'min {min(v) / 1e6:.4f}, max {max(v) / 1e6:.4f}\n')
    bj = json.load(open(os.path.join(src, 'bench_under_trace.json')))
    f.write(f"bench.py under the trace: roofline.launch_ms = {bj['roofline']['launch_ms']:.4f} ms over {bj['roofline']['launches_per_step']} launches per step, "
            f"valu_roofline.launch_ms_alone = {bj['valu_roofline']['launch_ms_alone']:.4f} ms\n")

# PMC passes (WSX_STREAMS=1: kernels one at a time), mean per launch and kernel
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in sorted(os.listdir(src)):
    p = os.path.join(src, d, 'p_counter_collection.csv')
    if not os.path.exists(p):
        continue
    for r in csv.DictReader(open(p)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        if k.startswith('at::') or k.startswith('__amd'):
            continue
        pmc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        dur[(k, r['Counter_Name'])].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
out = {}
for k, cs in pmc.items():
    out[k] = {c: {'mean_per_launch': sum(v) / len(v), 'launches': len(v), 'mean_launch_ms_under_pmc': sum(dur[(k, c)]) / len(v) / 1e6}
              for c, v in cs.items()}
bench = json.load(open(os.path.join(src, 'pmc3.json')))
out['_workload'] = bench['config']
json.dump(out, open(os.path.join(prof, f'{name}_pmc.json'), 'w'), indent=1)

fill = next(k for k in out if k.startswith('dtw_fill'))
n, T = bench['config']['reads_per_gpu'], bench['config']['samples_per_read']
fetch_kb, write_kb = out[fill]['FETCH_SIZE']['mean_per_launch'], out[fill]['WRITE_SIZE']['mean_per_launch']
traffic = {
    'kernel': fill, 'workload': {'reads': n, 'samples': T},
    'fetch_size_kb_per_launch': fetch_kb, 'write_size_kb_per_launch': write_kb,
    'correction': 'counters are KiB; FETCH and WRITE collected in separate --pmc passes; one launch = one pass over all reads '
                  '(WSX_STREAMS=1).  The x2 correction of MI355X_MICROARCH.md applies to 16-B-per-lane streaming reads; this kernel reads its '
                  'signal as 64-byte scalar loads plus 8-B-per-lane warm-up loads, a width the guide calls uncalibrated, so FETCH_SIZE is taken '
                  'as reported: it is 0.90 of the known 8 B x T x reads.  WRITE_SIZE matches the 16 B x T x reads of mask stores exactly.',
    'hbm_bytes_per_launch': (fetch_kb + write_kb) * 1024,
    'expected': '8 B x T x reads signal read + 16 B x T x reads back-pointer masks written (two 64-bit wave masks per row)',
}
json.dump(traffic, open(os.path.join(prof, f'{name}_traffic.json'), 'w'), indent=1)
rows_total = n * T
print(fill, 'VALU/row', out[fill]['SQ_INSTS_VALU']['mean_per_launch'] / rows_total, 'traffic GB', traffic['hbm_bytes_per_launch'] / 1e9)
for k in out:
    if k.startswith('_'): continue
    print(k, {c: round(v['mean_per_launch']) for c, v in out[k].items() if c in ('FETCH_SIZE', 'WRITE_SIZE')})
