"""Read a rocprofv3 --kernel-trace CSV and report, for the steady-state part of a bench run, how much of the wall time
has a DP fill kernel in flight, how many kernels overlap on average, and the per-kernel busy time.
Usage: python scripts/trace_overlap.py <kernel_trace.csv> [from_fraction to_fraction]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
upto = float(sys.argv[3]) if len(sys.argv) > 3 else 0.7
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows]
ev.sort()
t0, t1 = ev[0][0], max(e[1] for e in ev)
lo, hi = t0 + (t1 - t0) * skip, t0 + (t1 - t0) * upto
ev = [e for e in ev if e[0] >= lo and e[1] <= hi]
t0, t1 = ev[0][0], max(e[1] for e in ev)
wall = t1 - t0


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None or s > ce:
            if cs is not None:
                tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + (ce - cs if cs is not None else 0)


busy = defaultdict(int)
for s, e, k in ev:
    busy[k.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]] += e - s
fill = [(s, e) for s, e, k in ev if 'dtw_fill' in k]
print(f'window {wall/1e6:.2f} ms, {len(ev)} kernels')
print(f'any kernel in flight : {union([(s, e) for s, e, _ in ev]) / wall:.3f}')
print(f'fill in flight       : {union(fill) / wall:.3f}   sum(fill)/wall = {sum(e - s for s, e in fill) / wall:.3f}')
print(f'sum(all)/wall        : {sum(e - s for s, e, _ in ev) / wall:.3f}')
for k, v in sorted(busy.items(), key=lambda kv: -kv[1]):
    print(f'  {v/1e6:9.2f} ms  {v/wall:6.3f}  {k}')
