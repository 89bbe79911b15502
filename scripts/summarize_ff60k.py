"""Per-batch GPU time of the 60 000-read from_fast5 run from scripts/prof_from_fast5_60k.sh's traces (gpurun_out/prof_ff60k/):
kernels by name, the union of their intervals, the uploads, and one batch's sequence.  Usage: summarize_ff60k.py [dir]"""
import collections
import csv
import json
import re
import sys

d = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof_ff60k'
k = list(csv.DictReader(open(f'{d}/kernel_trace.csv')))
m = list(csv.DictReader(open(f'{d}/memory_copy_trace.csv')))
run = json.load(open(f'{d}/run.json'))['reader_sweep']
leg = run[[x for x in run if x.isdigit()][0]]


def nm(s):
    s = re.sub(r'\(anonymous namespace\)::', '', s)
    return re.sub(r'^void ', '', s).split('(')[0][:48]


z = [(int(x['Start_Timestamp']), int(x['End_Timestamp'])) for x in k if 'zstd_literals' in x['Kernel_Name']]
# the launches of the run itself: those of the bench's kernel leg (twelve, at the end, back to back) are dropped
gaps = [b[0] - a[0] for a, b in zip(z, z[1:])]
cut = max(range(len(gaps)), key=lambda i: gaps[i]) + 1 if gaps and max(gaps) > 5e8 else len(z)
z = z[:cut][-36:]
lo, hi = z[6][0], z[-3][0]
nb = sum(1 for a, _ in z if lo <= a < hi)
print(f'the run under the trace: {leg["reads_per_s"]:.0f} reads/s, wall {leg["wall_s"]:.3f} s, {leg["reader_processes"]} readers')
print(f'steady state: {nb} batches, one every {(hi - lo) / nb / 1e6:.2f} ms')
per, cnt, ivs = collections.Counter(), collections.Counter(), []
for x in k:
    a, b = int(x['Start_Timestamp']), int(x['End_Timestamp'])
    if lo <= a < hi:
        per[nm(x['Kernel_Name'])] += b - a
        cnt[nm(x['Kernel_Name'])] += 1
        ivs.append((a, b))
for n, v in per.most_common(14):
    print(f'  {v / nb / 1e6:7.3f} ms a batch, {cnt[n] / nb:5.1f} launches  {n}')
print(f'  {sum(per.values()) / nb / 1e6:7.3f} ms a batch: all kernels')
ivs.sort()
busy, (ca, cb) = 0, ivs[0]
for a, b in ivs[1:]:
    if a > cb:
        busy, ca, cb = busy + cb - ca, a, b
    else:
        cb = max(cb, b)
print(f'  kernels running {100 * (busy + cb - ca) / (hi - lo):.0f} % of the time')
cm = [(int(x['Start_Timestamp']), int(x['End_Timestamp'])) for x in m if lo <= int(x['Start_Timestamp']) < hi and x['Direction'].endswith('HOST_TO_DEVICE')]
print(f'  uploads: {sum(b - a for a, b in cm) / nb / 1e6:.2f} ms a batch in {len(cm) / nb:.0f} copies')
a0, a1 = z[12][0], z[13][0]
print('one batch (ms from its zstd_literals launch; events of 0.1 ms and more; * = an upload):')
ev = [(int(x['Start_Timestamp']), int(x['End_Timestamp']), nm(x['Kernel_Name'])) for x in k if a0 - 4e6 <= int(x['Start_Timestamp']) < a1]
ev += [(a, b, '* host to device') for a, b in cm if a0 - 4e6 <= a < a1]
for a, b, n in sorted(ev):
    if b - a >= 100000:
        print(f'  {(a - a0) / 1e6:8.3f} +{(b - a) / 1e6:6.3f}  {n}')
