"""Turn one scripts/profile_round.sh output directory into the files kept under profiles/.
Usage: summarize_profiles.py gpurun_out/<TAG> <name>   ->  profiles/<name>_kernel_stats.csv, <name>_pmc.json, <name>_traffic.json"""
import csv, json, os, sys, collections

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_hash  # noqa: E402  (the counters are valid for these sources only)

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
# per-kernel average above mixes the two, bench.py's roofline.fill_union_ms_per_launch is the first kind
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
    f.write(f"bench.py under the trace: roofline.fill_union_ms_per_launch = {bj['roofline'].get('fill_union_ms_per_launch', bj['roofline'].get('launch_ms')):.4f} ms over {bj['roofline']['launches_per_step']} launches per step, "
            f"valu_roofline.launch_ms_alone = {bj['valu_roofline']['launch_ms_alone']:.4f} ms, fill_union_ms_per_step = {bj['roofline']['fill_union_ms_per_step']:.4f}\n")

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

fills = [k for k in out if k.startswith('dtw_fill')]
n = bench['config']['reads_per_gpu']
samples_all = n * bench['config'].get('mean_samples_per_read', bench['config'].get('samples_per_read'))
waves_all = sum(out[k]['SQ_WAVES']['mean_per_launch'] for k in fills)
# profiles/fill_pmc.json: what bench.py quotes per fill kernel (valu_roofline, roofline.traffic); one entry per kernel name
idx_path = os.path.join(prof, 'fill_pmc.json')
idx = json.load(open(idx_path)) if os.path.exists(idx_path) else {}
traffic = {}
for fill in fills:
    f = out[fill]
    # a launch covers one pass over the reads of this kernel variant; with several variants in one workload the samples
    # are split by the variants' share of the reads (the loci draw their read lengths from the same range)
    samples = samples_all * f['SQ_WAVES']['mean_per_launch'] / waves_all
    fetch_kb, write_kb = f['FETCH_SIZE']['mean_per_launch'], f['WRITE_SIZE']['mean_per_launch']
    launch_s = f['GRBM_GUI_ACTIVE']['mean_launch_ms_under_pmc'] * 1e-3
    cyc = f['GRBM_GUI_ACTIVE']['mean_per_launch'] / 8.0  # the counter sums the 8 XCDs
    idx[fill] = {
        'valu_insts_per_wave_row': f['SQ_INSTS_VALU']['mean_per_launch'] / samples,   # one wave per read, one row per sample
        'lds_insts_per_wave_row': f['SQ_INSTS_LDS']['mean_per_launch'] / samples,
        'salu_insts_per_wave_row': f['SQ_INSTS_SALU']['mean_per_launch'] / samples,
        'clock_hz_observed': cyc / launch_s,
        'valu_busy': f['SQ_ACTIVE_INST_VALU']['mean_per_launch'] * 4.0 / (cyc * 1024),
        'lds_bank_conflict_cycles_per_wave_row': out[fill].get('SQ_LDS_BANK_CONFLICT', {}).get('mean_per_launch', float('nan')) / samples,
        'fetch_bytes_per_sample': fetch_kb * 1024 / samples, 'write_bytes_per_sample': write_kb * 1024 / samples,
        'hbm_bytes_per_sample': (fetch_kb + write_kb) * 1024 / samples,
        'launch_ms_under_pmc': f['GRBM_GUI_ACTIVE']['mean_launch_ms_under_pmc'], 'reads': f['SQ_WAVES']['mean_per_launch'],
        'samples_total': samples, 'source': f'profiles/{name}_pmc.json', 'workload': bench['config']['workload'],
        'kernel_source_hash': kernel_source_hash(),
    }
    traffic[fill] = {
        'workload': {'reads': f['SQ_WAVES']['mean_per_launch'], 'samples_total': samples},
        'fetch_size_kb_per_launch': fetch_kb, 'write_size_kb_per_launch': write_kb,
        'fetch_bytes_over_signal_bytes': fetch_kb * 1024 / (8.0 * samples),
        'write_bytes_per_sample': write_kb * 1024 / samples,
        'hbm_bytes_per_launch': (fetch_kb + write_kb) * 1024,
        'algorithmic_bytes_per_launch': 6.0 * samples + 16.0 * f['SQ_WAVES']['mean_per_launch'],
    }
    traffic[fill]['traffic_over_algorithmic'] = traffic[fill]['hbm_bytes_per_launch'] / traffic[fill]['algorithmic_bytes_per_launch']
    print(fill, json.dumps(idx[fill], indent=1))
traffic['_correction'] = ('counters are KiB; FETCH and WRITE collected in separate --pmc passes; one launch = one pass over all reads '
                          '(WSX_STREAMS=1).  The x2 correction of MI355X_MICROARCH.md applies to 16-B-per-lane streaming reads; the fill reads its '
                          'signal as 64-byte scalar loads plus 8-B-per-lane warm-up loads, a width the guide calls uncalibrated, so FETCH_SIZE is '
                          'taken as reported (fetch_bytes_over_signal_bytes = its ratio to the 8 B x samples the kernel must read).')
json.dump(traffic, open(os.path.join(prof, f'{name}_traffic.json'), 'w'), indent=1)
json.dump(idx, open(idx_path, 'w'), indent=1, sort_keys=True)
for k in out:
    if k.startswith('_'): continue
    print(k, {c: round(v['mean_per_launch']) for c, v in out[k].items() if c in ('FETCH_SIZE', 'WRITE_SIZE')})
