"""Bit-for-bit comparison of two builds of the library on the same seeded batches (every per-sample output and every result
record).  Usage: exp_bitcmp.py dump <out.npz>   (run once per build, WARPSTR_HIP_LIB selects it)
                 exp_bitcmp.py cmp <a.npz> <b.npz>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

if sys.argv[1] == 'cmp':
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = 0
    for k in a.files:
        same = a[k].tobytes() == b[k].tobytes()
        bad += not same
        print(f'{k:28s} {a[k].dtype} {a[k].shape} {"identical" if same else "DIFFERENT"}')
    sys.exit(1 if bad else 0)

from warpstr_amd import synth
from warpstr_amd.caller import HipCaller, pack_signals
out = {}
for name, pat, fl, T, n in [('hd', '(AGC)AACAGCCGCCAC(CGC)', 19, 2000, 100000), ('dm2', '((CAGG){CAGM})(CAGA)(CA)', 40, (500, 5000), 8000),
                            ('aaat', '(AAAT)', 110, (2271, 3701), 4000), ('agc', '(AGC)', 16, (600, 2500), 20000)]:
    locus = synth.make_locus(pat, fl, 11, max_states=64 if name == 'hd' else None)
    rng = np.random.default_rng(5)
    base = []
    for _ in range(256):
        rev = bool(rng.random() < 0.5)
        t = int(T) if np.isscalar(T) else int(rng.integers(T[0], T[1] + 1))
        hi = max(1, min(30, (t // 4 - 2 * fl - 12) // 14))
        base.append((synth.squiggle(locus, rev, t, rng, lo=1, hi=hi, sigma=0.0)[0], rev))
    pick = rng.integers(0, len(base), size=n)
    sig, off = pack_signals([base[i][0] for i in pick])
    sig = sig + rng.choice([0.15, 0.25, 0.4]) * rng.standard_normal(len(sig))
    aut = np.array([int(base[i][1]) for i in pick], dtype=np.int32)
    hip = HipCaller([locus.template, locus.reverse], [fl, fl], workspace_limit=64 << 30)
    res, ex = hip.call(sig, off, aut, want_traces=True, want_debug=True)
    out[name + '_results'] = res.view(np.uint8)
    for k in ('trace1', 'trace2', 'rescaled', 'badmask'):
        out[f'{name}_{k}'] = ex[k]
    print(name, 'called', int((res['status'] == 0).sum()), 'of', n, flush=True)
np.savez(sys.argv[2], **out)
