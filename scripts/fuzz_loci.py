"""Randomised parity sweep over LOCI: many seeded random locus patterns (nested units, optional blocks, IUPAC codes,
interruptions, flanks 12..150 -> automata of 30..320+ states, fan-in 2..4, every fill variant and the generic kernel), a
few dozen reads each, HIP caller vs the CPU oracle on all outputs incl. both state paths.  One line per kernel variant and a
final tally; exit code 1 on any mismatch.  (Test infrastructure: uses oracle/.)   Usage: fuzz_loci.py [n_loci] [reads_per_locus] [--configs | --smooth]
--smooth: rescaling.threshold in (1, 6] and the called automaton's levels perturbed per state (the signals follow the
unperturbed ones), so that many reads leave FITPACK's polynomial branch: knots added, smoothing iterated (fit_smooth_kernel);
the rescaled signal is compared bit for bit as well."""
import sys, os, time, collections
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle
from warpstr_amd import synth
from warpstr_amd.caller import HipCaller, pack_signals

# caller / rescaler settings a locus may be called with (config.py:91-119 upstream); `configs=True` cycles through them
CONFIGS = [dict(), dict(min_values_per_state=3), dict(min_values_per_state=5), dict(min_values_per_state=2),
           dict(method='median'), dict(reps_as_one=True), dict(states_in_segment=4), dict(min_values_per_state=3, method='median'),
           dict(threshold=0.8, max_std=0.3)]


SMOOTH = [dict(threshold=1.5, max_std=3.0), dict(threshold=3.0, max_std=3.0), dict(threshold=6.0, max_std=3.0),
          dict(threshold=6.0, max_std=3.0, min_values_per_state=3), dict(threshold=4.0, max_std=2.0, method='median')]


def run(n_loci, per, seed=2026, verbose=True, configs=False, smooth=False):
    """-> (reads compared, mismatches, {kernel name: reads})"""
    from warpstr_amd.caller import CallerConfig, RescalerConfig
    units = ['AGC', 'AAAT', 'GGCCCC', 'CAG', 'CTG', 'CCTG', 'NGC', 'RY', 'CAGM', 'AAGGG', 'GAA', 'TTTTA', 'GCN', 'CGG', 'AT', 'ATTCT']
    rng = np.random.default_rng(seed)
    oracle.lib()
    tot = bad = skipped = n_smooth = 0
    import copy
    by_kernel = collections.Counter()
    bad_by_kernel = collections.Counter()
    t00 = time.time()
    for li in range(n_loci):
        pat = ''
        for _u in range(int(rng.integers(1, 4))):
            unit = units[int(rng.integers(len(units)))]
            if rng.random() < 0.2:
                unit = '(' + unit + '){' + units[int(rng.integers(len(units)))] + '}'
            pat += '(' + unit + ')'
            if rng.random() < 0.4:
                pat += ''.join('ACGT'[i] for i in rng.integers(0, 4, size=int(rng.integers(1, 14))))
        fl = int(rng.integers(12, 150))
        try:
            locus = synth.make_locus(pat, fl, int(rng.integers(1_000_000)))
        except Exception:  # (patterns the automaton compiler refuses, as upstream does)
            skipped += 1
            continue
        S = max(locus.template.n_states, locus.reverse.n_states)
        cfg = CONFIGS[li % len(CONFIGS)] if configs else (SMOOTH[li % len(SMOOTH)] if smooth else {})
        called = locus
        if smooth:  # the automaton the reads are called against: every level off by its own amount
            called = copy.deepcopy(locus)
            spread = float(rng.choice([0.8, 1.5, 2.5]))
            for tb in (called.template, called.reverse):
                tb.value = (tb.value + rng.normal(0.0, spread, size=tb.n_states)).astype(np.float64)
        cc = CallerConfig(**{k: v for k, v in cfg.items() if k in ('min_values_per_state', 'states_in_segment')})
        rc = RescalerConfig(**{k: v for k, v in cfg.items() if k in ('method', 'reps_as_one', 'threshold', 'max_std')})
        prm = oracle.Params(min_values_per_state=cc.min_values_per_state, states_in_segment=cc.states_in_segment,
                            threshold=rc.threshold, max_std=rc.max_std, method=rc.method, reps_as_one=rc.reps_as_one)
        try:
            hip = HipCaller([called.template, called.reverse], [fl, fl], caller_config=cc, rescaler_config=rc)
        except Exception as e:
            skipped += 1
            if verbose:
                print(f'{pat} fl={fl} S={S}: not accepted by wsx_caller_create ({str(e)[:80]})', flush=True)
            continue
        sigs, revs = [], []
        for i in range(per):
            rev = bool(rng.random() < 0.5)
            t = int(rng.integers(max(6 * S, 400), max(6 * S, 400) + 2500))
            hi = max(2, min(30, (t // 4 - 2 * fl - 12) // 14))
            s, _ = synth.squiggle(locus, rev, t, rng, lo=1, hi=hi, sigma=float(rng.choice([0.15, 0.25, 0.4])))
            sigs.append(s)
            revs.append(rev)
        sig, off = pack_signals(sigs)
        aut = np.array(revs, dtype=np.int32)
        res, ex = hip.call(sig, off, aut, want_traces=True, want_debug=smooth)
        oa = [oracle.Automaton.from_table(called.template, fl), oracle.Automaton.from_table(called.reverse, fl)]
        knots = [0] * per

        def check(i):
            o = oracle.call_read(oa[aut[i]], sigs[i], prm)
            if int(res['status'][i]) != o.status:
                return f'read {i}: status {res["status"][i]} vs {o.status}'
            if o.status:
                return None
            knots[i] = o.fit_knots
            sl = slice(off[i], off[i + 1])
            if smooth and not np.array_equal(ex['rescaled'][sl], o.rescaled):
                return f'read {i}: rescaled ({o.fit_knots} knots)'
            if not np.array_equal(ex['trace1'][sl], o.trace1):
                return f'read {i}: trace1'
            if not np.array_equal(ex['trace2'][sl], o.trace2):
                return f'read {i}: trace2'
            if (res['len1'][i], res['len2'][i], res['n_trans1'][i], res['n_trans2'][i]) != (o.len1, o.len2, o.n_trans1, o.n_trans2):
                return f'read {i}: lengths'
            for a, b in ((res['cost1'][i], o.cost1), (res['cost2'][i], o.cost2), (res['dtw_end_cost1'][i], o.dtw_end_cost1),
                         (res['dtw_end_cost2'][i], o.dtw_end_cost2)):
                if not (a == b or (np.isnan(a) and np.isnan(b)) or abs(a - b) <= 1e-9 * max(abs(a), abs(b))):
                    return f'read {i}: cost {a} vs {b}'
            return None
        with ThreadPoolExecutor(os.cpu_count()) as pool:
            errs = [e for e in pool.map(check, range(per)) if e]
        # WarpSTR.warp on its own (wsx_warp_batch): a random bad-repeat mask on half of the reads, the whole last DP row
        nw = min(per, 8)
        wmask = np.zeros(int(off[nw]), np.uint8)
        for i in range(0, nw, 2):
            a0 = int(off[i] + rng.integers(0, len(sigs[i]) // 2))
            wmask[a0:a0 + int(rng.integers(1, len(sigs[i]) // 2))] = 1
        wr = hip.warp(sig[:off[nw]], off[:nw + 1], aut[:nw], mask=wmask, want_last_row=True)
        m_ = cc.min_values_per_state
        for i in range(nw):
            sl = slice(int(off[i]), int(off[i + 1]))
            a_ = oa[aut[i]]
            try:
                D = oracle.dtw_fill(a_, sigs[i], wmask[sl], m_)
                otr = oracle.backtrack(a_, D, sigs[i], wmask[sl], m_)
            except RuntimeError:
                if wr['status'][i] == 0:
                    errs.append(f'warp read {i}: oracle refused, status 0')
                continue
            lr = wr['last_row'][i, :a_.n_states]
            same = (lr == D[-1]) | (np.isinf(lr) & np.isinf(D[-1]))
            if wr['status'][i] != 0 or not same.all() or not np.array_equal(wr['trace'][sl], otr):
                errs.append(f'warp read {i}: status {wr["status"][i]}, last row equal {bool(same.all())}, '
                            f'trace equal {np.array_equal(wr["trace"][sl], otr)}')
        for k in (0, 1):
            by_kernel[hip.kernel_name(k) + (' ' + str(cfg) if cfg else '')] += int((aut == k).sum())
        tot += per
        bad += len(errs)
        n_smooth += sum(k > 8 for k in knots)
        if errs:
            bad_by_kernel[hip.kernel_name(0) + (' ' + str(cfg) if cfg else '')] += len(errs)
            print(f'MISMATCH {pat} fl={fl} S={locus.template.n_states}/{locus.reverse.n_states} {hip.kernel_name(0)} / '
                  f'{hip.kernel_name(1)}: {errs[:3]}', flush=True)
        if verbose and li % 100 == 99:
            print(f'... {li + 1} loci, {tot} reads, {bad} mismatches, {time.time() - t00:.0f} s', flush=True)
        hip.close()
    if verbose:
        for k, n in sorted(by_kernel.items()):
            print(f'{k:40s} {n:7d} reads  {bad_by_kernel.get(k, 0)} mismatching')
        print(f'TOTAL {n_loci - skipped} loci ({skipped} skipped), {tot} reads, {bad} mismatches, {time.time()-t00:.0f} s'
              + (f'; {n_smooth} reads through FITPACK\'s smoothing branch' if smooth else ''))
    return tot, bad, dict(by_kernel)


if __name__ == '__main__':
    n_loci = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    sys.exit(1 if run(n_loci, per, configs='--configs' in sys.argv, smooth='--smooth' in sys.argv)[1] else 0)
