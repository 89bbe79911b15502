"""Randomised sweep of the signal loader: wsx_prepare_signals (spike removal, whole-read MAD normalisation, slice) against
the host restatement warpstr_amd/signal_prep.py (itself pinned by the vector recorded from upstream's Fast5 code) on seeded
raw reads: short reads (the one-wavefront-per-read kernel), long reads (the general kernels), wide value ranges, runs of
outliers, constant stretches, segments that start or end outside the read.  Bit-exact or it counts as a mismatch.
Usage: fuzz_loader.py [n_batches]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from warpstr_amd import synth
from warpstr_amd.caller import HipCaller
from warpstr_amd.signal_prep import process_raw

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 50
locus = synth.make_locus('(AGC)', 16, 5)
hip = HipCaller([locus.template, locus.reverse], [16, 16])
rng = np.random.default_rng(77)
tot = bad = 0
kinds = {}
t0 = time.time()
for b in range(nb):
    raws, pos, kind = [], [], []
    for k in range(64):
        c = int(rng.integers(0, 8))
        if c <= 2:    # segment-sized reads
            n = int(rng.integers(5, 8192)); name = 'short'
        elif c <= 4:
            n = int(rng.integers(8192, 60000)); name = 'medium'
        elif c == 5:
            n = int(rng.integers(60000, 400000)); name = 'long'
        else:
            n = int(rng.integers(50, 3000)); name = 'short-odd'
        mu, sd = float(rng.uniform(300, 900)), float(rng.uniform(5, 150))
        raw = rng.normal(mu, sd, size=n)
        if name == 'short-odd':
            m = int(rng.integers(0, 4))
            if m == 0: raw[:] = np.round(mu)                                   # constant read
            elif m == 1: raw[rng.random(n) < 0.5] = mu + 3000                  # bimodal, wide
            elif m == 2: raw = np.repeat(rng.normal(mu, sd, size=n // 7 + 1), 7)[:n]  # plateaus (ties)
            else: raw[::2] = -2000                                             # negative values
        raw = np.clip(raw, -32768, 32767).astype(np.int16)
        nsp = int(rng.integers(0, max(2, n // 100)))
        if nsp:
            idx = rng.integers(0, n, size=nsp)
            raw[idx] = rng.choice([0, 30, 100, 1200, 2500, 4000, -500], size=nsp)
        if rng.random() < 0.3 and n > 60:                                      # runs of adjacent outliers, also at the ends
            a = int(rng.integers(0, n - 10)); raw[a:a + int(rng.integers(2, 9))] = 3000
            raw[:int(rng.integers(0, 4))] = 2500
            raw[n - int(rng.integers(0, 3)):] = 5
        raws.append(raw)
        a = int(rng.integers(0, n))
        pos.append((a, int(rng.integers(a, n + 40))))
        kind.append(name)
    for mode in ('Brute', 'None'):
        out, ooff, _ = hip.prepare_signals(raws, pos, mode)
        for i, (raw, p) in enumerate(zip(raws, pos)):
            ref = process_raw(raw, p, mode)
            got = out[ooff[i]:ooff[i + 1]]
            ok = len(got) == len(ref) and np.array_equal(got, ref, equal_nan=True)
            tot += 1
            kinds[(kind[i], mode)] = kinds.get((kind[i], mode), 0) + 1
            if not ok:
                bad += 1
                print(f'MISMATCH batch {b} read {i} ({kind[i]}, n={len(raw)}, segment {p}, {mode})', flush=True)
    if b % 10 == 9:
        print(f'... {b + 1} batches, {tot} reads, {bad} mismatches, {time.time() - t0:.0f} s', flush=True)
for k, v in sorted(kinds.items()):
    print(f'{k[0]:10s} {k[1]:6s} {v:6d} reads')
print(f'TOTAL {tot} read preparations, {bad} mismatches, {time.time() - t0:.0f} s')
sys.exit(1 if bad else 0)
