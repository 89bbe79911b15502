"""Randomised parity sweep: HIP caller vs the CPU oracle on many seeded reads (all outputs, incl. both state paths).
Prints one line per locus and a final tally; exit code 1 on any mismatch.  (Test infrastructure: uses oracle/.)"""
import sys, os, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle
from warpstr_amd import synth
from warpstr_amd.caller import HipCaller, pack_signals

SPECS = [('(AGC)', 16, (600, 2500), 6000), ('(AGC)AACAGCCGCCAC(CGC)', 19, 2000, 8000), ('(AAAT)', 110, (2271, 3701), 1500),
         ('((CAGG){CAGM})(CAGA)(CA)', 40, (900, 5000), 3000), ('(NGC)', 24, (900, 2200), 3000), ('(CTG)', 30, 1500, 4000),
         ('(GGCCCC)', 25, (1200, 3000), 3000), ('(CCTG)(TG)', 20, (800, 2400), 3000)]
n_arg = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
oracle.lib()
tot = bad = 0
t00 = time.time()
for li, (pat, fl, T, n) in enumerate(SPECS):
    n = max(64, int(n * n_arg))
    locus = synth.make_locus(pat, fl, 7000 + li)
    rng = np.random.default_rng(100 + li)
    sigs, revs = [], []
    for i in range(n):
        rev = bool(rng.random() < 0.5)
        t = int(T) if np.isscalar(T) else int(rng.integers(T[0], T[1] + 1))
        hi = max(2, min(30, (t // 4 - 2 * fl - 12) // 14))
        s, _ = synth.squiggle(locus, rev, t, rng, lo=1, hi=hi, sigma=float(rng.choice([0.15, 0.25, 0.4])))
        sigs.append(s); revs.append(rev)
    sig, off = pack_signals(sigs)
    aut = np.array(revs, dtype=np.int32)
    hip = HipCaller([locus.template, locus.reverse], [fl, fl])
    res, ex = hip.call(sig, off, aut, want_traces=True)
    oa = [oracle.Automaton.from_table(locus.template, fl), oracle.Automaton.from_table(locus.reverse, fl)]
    def check(i):
        o = oracle.call_read(oa[aut[i]], sigs[i])
        if int(res['status'][i]) != o.status: return f'read {i}: status {res["status"][i]} vs {o.status}'
        if o.status: return None
        sl = slice(off[i], off[i + 1])
        if not np.array_equal(ex['trace1'][sl], o.trace1): return f'read {i}: trace1'
        if not np.array_equal(ex['trace2'][sl], o.trace2): return f'read {i}: trace2'
        if (res['len1'][i], res['len2'][i], res['n_trans1'][i], res['n_trans2'][i]) != (o.len1, o.len2, o.n_trans1, o.n_trans2):
            return f'read {i}: lengths'
        for a, b in ((res['cost1'][i], o.cost1), (res['cost2'][i], o.cost2), (res['dtw_end_cost1'][i], o.dtw_end_cost1),
                     (res['dtw_end_cost2'][i], o.dtw_end_cost2)):
            if not (a == b or (np.isnan(a) and np.isnan(b)) or abs(a - b) <= 1e-9 * max(abs(a), abs(b))): return f'read {i}: cost {a} vs {b}'
        return None
    with ThreadPoolExecutor(os.cpu_count()) as pool:
        errs = [e for e in pool.map(check, range(n)) if e]
    ok = int((res['status'] == 0).sum())
    tot += n; bad += len(errs)
    print(f'{pat:28s} fl={fl:3d} S={locus.template.n_states}/{locus.reverse.n_states} {hip.kernel_name(0)}: {n} reads, '
          f'{ok} called, {len(errs)} mismatches {errs[:3]}', flush=True)
print(f'TOTAL {tot} reads, {bad} mismatches, {time.time()-t00:.0f} s')
sys.exit(1 if bad else 0)
