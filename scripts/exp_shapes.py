"""Experiment: fill-kernel time and whole-call time for the BASELINE config shapes (host-memory API)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from warpstr_amd import synth
from warpstr_amd.caller import HipCaller, pack_signals

def run(name, pattern, fl, T, n, seed=1, max_states=None):
    locus = synth.make_locus(pattern, fl, seed, max_states=max_states)
    rng = np.random.default_rng(seed)
    base = []
    for _ in range(min(n, 128)):
        rev = bool(rng.random() < 0.5)
        t = int(T) if np.isscalar(T) else int(rng.integers(T[0], T[1] + 1))
        hi = max(1, min(30, (t // 4 - 2 * fl - 12) // 14))
        base.append((synth.squiggle(locus, rev, t, rng, lo=1, hi=hi, sigma=0.0)[0], rev))
    sigs, revs = [], []
    for i in range(n):
        s, rev = base[i % len(base)]
        sigs.append(s + 0.25 * rng.normal(size=len(s)))
        revs.append(rev)
    sig, off = pack_signals(sigs)
    aut = np.array(revs, dtype=np.int32)
    hip = HipCaller([locus.template, locus.reverse], [fl, fl])
    hip.set_streams(1)  # kernels one at a time: per-kernel durations are not inflated by overlap
    for _ in range(2):
        t0 = time.perf_counter()
        res, _ = hip.call(sig, off, aut)
        dt = time.perf_counter() - t0
        tm = hip.last_timing()
    cells = 2 * sum(len(s) * (locus.reverse.n_states if r else locus.template.n_states) for s, r in zip(sigs, revs))
    ok = int((res['status'] == 0).sum())
    print(f'{name:8s} S={locus.template.n_states}/{locus.reverse.n_states} {hip.kernel_name(0)} n={n} ok={ok} '
          f'fill={tm["dp_kernel_ms"]:.2f}ms device={tm["total_ms"]:.2f}ms host_call={dt*1e3:.1f}ms '
          f'fill_cells/s={cells/tm["dp_kernel_ms"]/1e-3:.3g} reads/s(device)={n/tm["total_ms"]/1e-3:.3g}', flush=True)

if len(sys.argv) > 1 and sys.argv[1] == 'cfg3':  # just the headline shape (fill-kernel experiments)
    run('cfg3', '(AGC)AACAGCCGCCAC(CGC)', 19, 2000, int(sys.argv[2]) if len(sys.argv) > 2 else 50000, seed=2024, max_states=64)
    sys.exit(0)
run('cfg2', '(AGC)', 16, 1500, 20000)
run('cfg3', '(AGC)AACAGCCGCCAC(CGC)', 19, 2000, 20000, seed=2024, max_states=64)
run('cfg5', '((CAGG){CAGM})(CAGA)(CA)', 40, (500, 5000), 8000)
run('cfg1', '(AAAT)', 110, (2271, 3701), 4000)
run('ngc', '(NGC)', 24, 1800, 8000)
