"""Per-kernel totals of a rocprofv3 --kernel-trace CSV, per bench step.  Usage: kstats.py <kernel_trace.csv> <n_steps>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot, cnt = defaultdict(int), defaultdict(int)
for r in rows:
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    if k.startswith('at::') or k.startswith('__amd'):
        continue
    tot[k] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    cnt[k] += 1
allt = sum(tot.values())
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f'{v/1e6/steps:8.3f} ms/step  {cnt[k]/steps:6.1f} launches/step  {k}')
print(f'{allt/1e6/steps:8.3f} ms/step total')
