"""Experiment: fill kernel time, unmasked vs masked(all-zero mask) vs masked(real-ish mask), same signals."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from warpstr_amd import synth
from warpstr_amd.caller import HipCaller, pack_signals

locus = synth.make_locus('(AGC)AACAGCCGCCAC(CGC)', 19, 2024, max_states=64)
n, T = 20000, 2000
rng = np.random.default_rng(0)
tpl = [synth.squiggle(locus, False, T, rng, sigma=0.0)[0] for _ in range(256)]
sig = np.stack([tpl[i % 256] for i in range(n)]) + 0.25 * rng.normal(size=(n, T))
sig = sig.reshape(-1)
off = np.arange(n + 1, dtype=np.int64) * T
aut = np.zeros(n, np.int32)
hip = HipCaller([locus.template, locus.reverse], [19, 19])
for name, mask in [('unmasked', None), ('zero mask', np.zeros(n * T, np.uint8)),
                   ('10% mask', (rng.random(n * T) < 0.1).astype(np.uint8)),
                   ('block mask', np.tile((np.arange(T) % 400 < 50).astype(np.uint8), n))]:
    for rep in range(3):
        out = hip.warp(sig, off, aut, mask=mask)
        tm = hip.last_timing()
    print(f'{name:12s} fill {tm["dp_kernel_ms"]:.3f} ms  launches {tm["dp_launches"]}', flush=True)
