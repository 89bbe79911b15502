#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the upstream caller.

DEVELOPMENT-CONTAINER ONLY (needs /root/reference; see _ref_import.py).  Usage:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/generate_golden.py            # all default-config cases
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/generate_golden.py --alt      # alternative-config cases

Inputs are produced by this repository's own seeded generator (warpstr_amd/synth.py); every
expected value below is computed by the reference's own functions:
  StateAutomata (src/caller/automata.py:43-48), WarpSTR._calc_dtw_astates / _backtracking
  (src/caller/caller.py:198-301), WarpResult.create_alignment (65-96), rescale_signal (304-313),
  mask_bad_repeats (330-339), WarpSTR.run (117-149), CallerWrapper.break_into_units /
  collapse_repeats / reverse_uniq_sequence (src/caller/wrapper.py:78-84,162-248),
  normalize_signal_mad / Fast5.brute_remove (src/schemas/fast5.py:90-114), CallerWrapper.check_high_similarity
  (src/caller/wrapper.py:122-160).
The files written are data only (arrays and strings): no reference source text is stored.
"""
import argparse
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from _ref_import import import_reference  # noqa: E402

from warpstr_amd import synth  # noqa: E402

CASES = [
    # name, pattern, flank_length, flank seed, T, reads, read seed, (lo, hi)
    ('agc_fl16', '(AGC)', 16, 11, 1500, 6, 101, (5, 30)),
    ('agc_fl29', '(AGC)', 29, 12, 2000, 4, 102, (5, 30)),
    ('aaat_fl110', '(AAAT)', 110, 13, 2600, 4, 103, (5, 30)),
    ('hd_fl20', '(AGC)AACAGCCGCCAC(CGC)', 20, 14, 2000, 6, 104, (5, 30)),
    ('dm2_fl40', '((CAGG){CAGM})(CAGA)(CA)', 40, 15, 3000, 4, 105, (5, 20)),
    ('ngc_fl20', '(NGC)', 20, 16, 1500, 4, 106, (5, 30)),
    ('agc_fl16_ragged', '(AGC)', 16, 17, (600, 2500), 6, 107, (3, 25)),
]
ALT_CASES = [
    # name, pattern, fl, flank seed, T, reads, read seed, (lo,hi), config overrides
    ('alt_m3_median', '(AGC)', 20, 21, 1500, 4, 201, (5, 30),
     dict(min_values_per_state=3, method='median', states_in_segment=5)),
    ('alt_repsasone', '(AGC)', 20, 22, 1500, 4, 202, (5, 30), dict(reps_as_one=True, max_std=0.6, threshold=0.4)),
    # rescaling.threshold above 1 (src/config.py:97-100 allows any positive value): states up to 1.5 normalised units off
    # their level enter the spline fit; FITPACK still stays on its polynomial branch (ier = -2) for these reads
    ('alt_thr15', '(AGC)AACAGCCGCCAC(CGC)', 20, 23, 1700, 6, 203, (6, 20), dict(threshold=1.5, max_std=0.9)),
]


def run_reference_read(ns, sta, flank_length, reverse, signal, full_matrix=False):
    """Step through WarpSTR.run (src/caller/caller.py:117-149) keeping every intermediate."""
    C = ns.caller
    w = C.WarpSTR(flank_length, sta.states, sta.endstate, sta.mask, None, reverse, 'r')
    out = {}
    mask0 = np.full(len(signal), False)
    D1 = w._calc_dtw_astates(signal, sta.states, mask0)
    tr1 = w._backtracking(D1, sta.states, signal, mask0)
    wr1 = C.WarpResult(tr1)
    al1 = wr1.create_alignment(sta.states, signal)
    resc = C.rescale_signal(signal, al1)
    start, end, badmask = C.mask_bad_repeats(signal, sta.mask, wr1.trace, wr1.state_transitions)
    mask1 = np.asarray(badmask, dtype=bool) if badmask else np.full(len(signal), False)
    D2 = w._calc_dtw_astates(resc, sta.states, mask1)
    tr2 = w._backtracking(D2, sta.states, resc, mask1)
    wr2 = C.WarpResult(tr2)
    al2 = wr2.create_alignment(sta.states, resc)
    resc2 = C.rescale_signal(resc, al2)
    rstart, rend, _ = C.mask_bad_repeats(resc2, sta.mask, wr2.trace, wr2.state_transitions)
    cost = np.mean([a.cost for a in al1[start:end]])
    rcost = np.mean([a.cost for a in al2[rstart:rend]])
    res = w.run(signal)  # the real thing, to make sure the stepping above is faithful
    assert res.seq == w._get_sequence(reverse, wr1) and res.resc_seq == w._get_sequence(reverse, wr2)
    assert (res.cost == cost or (np.isnan(res.cost) and np.isnan(cost))) and \
        (res.resc_cost == rcost or (np.isnan(res.resc_cost) and np.isnan(rcost)))
    out.update(
        trace1=tr1.astype(np.int32), trace2=tr2.astype(np.int32),
        dlast1=D1[-1].copy(), dlast2=D2[-1].copy(),
        dsum1=np.array([np.sum(D1[np.isfinite(D1)]), np.count_nonzero(np.isfinite(D1))]),
        dsum2=np.array([np.sum(D2[np.isfinite(D2)]), np.count_nonzero(np.isfinite(D2))]),
        align1_value=np.array([a.state_value for a in al1]), align1_expected=np.array([a.expected for a in al1]),
        align1_good=np.array([a.good_enough() for a in al1], dtype=np.uint8),
        align2_value=np.array([a.state_value for a in al2]), align2_expected=np.array([a.expected for a in al2]),
        align2_good=np.array([a.good_enough() for a in al2], dtype=np.uint8),
        rescaled=np.asarray(resc, dtype=np.float64), rescaled2=np.asarray(resc2, dtype=np.float64),
        badmask=np.asarray(badmask, dtype=np.uint8),
        idx=np.array([start, end, rstart, rend], dtype=np.int64),
        cost=np.array([cost, rcost]),
        seq=np.array([res.seq, res.resc_seq]),
    )
    if full_matrix:
        out['D1'] = D1
        out['D2'] = D2
    return out


def gen_case(ns, name, pattern, fl, fseed, T, n_reads, rseed, lohi, outdir):
    locus = synth.make_locus(pattern, fl, fseed)
    sigs, revs, truth = synth.batch(locus, n_reads, T, rseed, lo=lohi[0], hi=lohi[1])
    write_case(ns, name, pattern, fl, locus, sigs, revs, truth, outdir)


def write_case(ns, name, pattern, fl, locus, sigs, revs, truth, outdir, extra=None):
    n_reads = len(sigs)
    # make sure both strands are present
    tmp_seq = locus.left_t + pattern + locus.right_t
    rev_seq = locus.left_r + ns.wrapper.CallerWrapper.reverse_uniq_sequence(pattern) + locus.right_r
    stas = {False: ns.automata.StateAutomata(tmp_seq), True: ns.automata.StateAutomata(rev_seq)}
    data = {'pattern': np.array(pattern), 'flank_length': np.array(fl),
            'flanks': np.array([locus.left_t, locus.right_t, locus.left_r, locus.right_r]),
            'n_reads': np.array(n_reads), 'reverse': np.array(revs, dtype=np.uint8)}
    for tag, sta in (('t', stas[False]), ('r', stas[True])):
        data[f'{tag}_value'] = np.array([s.value for s in sta.states])
        data[f'{tag}_seq_idx'] = np.array([s.seq_idx for s in sta.states], dtype=np.int32)
        ptr = np.cumsum([0] + [len(s.incoming) for s in sta.states]).astype(np.int32)
        data[f'{tag}_pred_ptr'] = ptr
        data[f'{tag}_pred_idx'] = np.array([p.idx for s in sta.states for p in s.incoming], dtype=np.int32)
        data[f'{tag}_mask'] = np.array(sta.mask, dtype=np.uint8)
        data[f'{tag}_endstate'] = np.array(sta.endstate)
        data[f'{tag}_kmers'] = np.array([s.kmer for s in sta.states])
    for i, (sig, rev) in enumerate(zip(sigs, revs)):
        r = run_reference_read(ns, stas[rev], fl, rev, sig, full_matrix=(name == 'agc_fl16' and i == 0))
        data[f'r{i}_signal'] = sig
        data[f'r{i}_truth'] = np.array(truth[i])
        for k, v in r.items():
            data[f'r{i}_{k}'] = v
        print(f'  {name} read {i} rev={int(rev)} T={len(sig)} S={len(stas[rev].states)} truth={truth[i]} '
              f'len1={len(r["seq"][0])} len2={len(r["seq"][1])} cost={r["cost"]}')
    data.update(extra or {})
    np.savez_compressed(os.path.join(outdir, f'{name}.npz'), **data)


def gen_real(ns, outdir):
    """The upstream test case (README.md section 2, test/test_caller_only/example.csv: 10 real R9.4 reads of an (AAAT)
    locus).  Raw samples come out of the upstream multi-read fast5 through this repository's reader (h5py and the VBZ
    plugin are absent offline); spike removal, normalisation and the caller are the reference's own functions.  GRCh38
    is not available, so the flanks are the pile-up consensus written by real/make_flanks.py.  The fast5 and the csv are
    copied beside the vectors as data fixtures for the loader tests."""
    import csv
    import shutil
    from types import SimpleNamespace
    from warpstr_amd.fast5 import Fast5File
    real = os.path.join(outdir, 'real')
    with open(os.path.join(real, 'flanks.json')) as f:
        fj = json.load(f)
    ref_root = '/root/reference'
    rows = list(csv.DictReader(open(os.path.join(ref_root, 'test/test_caller_only/example.csv'))))
    shutil.copyfile(os.path.join(ref_root, 'test/test_caller_only/example.csv'), os.path.join(real, 'example.csv'))
    shutil.copyfile(os.path.join(ref_root, rows[0]['fast5_path']), os.path.join(real, 'batch_0.fast5'))
    locus = SimpleNamespace(left_t=fj['left_template'], right_t=fj['right_template'], left_r=fj['left_reverse'],
                            right_r=fj['right_reverse'])
    sigs, revs, names, pos = [], [], [], []
    with Fast5File(os.path.join(ref_root, rows[0]['fast5_path'])) as f5:
        for row in rows:
            raw = f5.raw_signal(row['read_name'])
            norm = ns.normalize_signal_mad(ns.Fast5.brute_remove(raw))  # fast5.py:45-57 with spike_removal = Brute
            lo, hi = int(row['l_start_raw']), int(row['r_end_raw'])
            sigs.append(np.asarray(norm[lo:hi + 1], dtype=np.float64))
            revs.append(row['reverse'].upper() == 'TRUE')
            names.append(row['read_name'])
            pos.append((lo, hi, len(raw)))
    extra = {'names': np.array(names), 'raw_span': np.array(pos, dtype=np.int64)}
    write_case(ns, 'real_aaat', fj['sequence'], fj['flank_length'], locus, sigs, revs, [[-1]] * len(sigs), outdir, extra)


def gen_units(ns, outdir):
    """Host-side string helpers (src/caller/wrapper.py:78-84,162-248) and signal pre-processing vectors."""
    W = ns.wrapper.CallerWrapper
    # NB: a top-level '{..}' block makes upstream's collapse_repeats spin forever (an empty repeat unit always
    # matches, src/caller/wrapper.py:231-247), so no such pattern is listed here.
    pats = ['(AGC)', '(AAAT)', '(AGC)AACAGCCGCCAC(CGC)', '((CAGG){CAGM})(CAGA)(CA)', '(CCTG)(TG)', '(NGC)',
            '(ARC)TTGGA(CCG)', 'AC(GGCCCC)T']
    out = {}
    for p in pats:
        stub = W.__new__(W)
        units, repeat_units, offsets = W.break_into_units(stub, p)
        stub.repeat_units, stub.offsets = repeat_units, offsets
        seqs = []
        rng = np.random.default_rng(5)
        for _ in range(6):
            s, _c = synth.instantiate_pattern(p, rng, 2, 9)
            seqs.append(s)
        seqs.append('ACGT')
        out[p] = dict(reverse=W.reverse_uniq_sequence(p), units=units, repeat_units=repeat_units, offsets=offsets,
                      collapse=[[s, W.collapse_repeats(stub, s)] for s in seqs])
    with open(os.path.join(outdir, 'units.json'), 'w') as f:
        json.dump(out, f, indent=1)

    rng = np.random.default_rng(9)
    raw = rng.normal(500, 60, size=4000).astype(np.int16)
    raw[[1, 2, 3, 50, 51, 700, 3998]] = [1500, 100, 1200, 30, 2000, 1001, 249]
    cleaned = ns.Fast5.brute_remove(raw)
    norm = ns.normalize_signal_mad(cleaned)
    np.savez_compressed(os.path.join(outdir, 'signal_prep.npz'), raw=raw, cleaned=cleaned, norm=norm,
                        pore_level_norm=np.asarray(ns.pore_model.table['level_norm'].values))


def gen_negative(ns, outdir):
    """Reference failure modes: recorded as facts, not emulated."""
    out = {}
    # (a) flank_length < 16: IndexError in find_event_borders (src/caller/caller.py:395-397) on some reads
    locus = synth.make_locus('(AGC)', 14, 31)
    sigs, revs, _ = synth.batch(locus, 8, 1200, 301)
    tmp_seq = locus.left_t + '(AGC)' + locus.right_t
    rev_seq = locus.left_r + '(GCT)' + locus.right_r
    stas = {False: ns.automata.StateAutomata(tmp_seq), True: ns.automata.StateAutomata(rev_seq)}
    data = {'flanks': np.array([locus.left_t, locus.right_t, locus.left_r, locus.right_r]),
            'reverse': np.array(revs, dtype=np.uint8)}
    outcome = []
    for i, (sig, rev) in enumerate(zip(sigs, revs)):
        data[f'r{i}_signal'] = sig
        w = ns.caller.WarpSTR(14, stas[rev].states, stas[rev].endstate, stas[rev].mask, None, rev, 'r')
        try:
            res = w.run(sig)
            outcome.append(f'ok:{len(res.seq)}:{len(res.resc_seq)}')
        except Exception as e:  # noqa: BLE001
            outcome.append(type(e).__name__)
    data['outcome'] = np.array(outcome)
    print('  fl14 outcomes', outcome)
    np.savez_compressed(os.path.join(outdir, 'neg_fl14.npz'), **data)
    out['fl14'] = outcome
    return out


SIMILARITY_SEQUENCES = ['(AGC)', '(AAAT)', '(AGC)AACAGCCGCCAC(CGC)', '((CAGG){CAGM})(CAGA)(CA)', '(NGC)', '(GGCCCC)', '(CAG)(CAA)(CAG)',
                        '(A)', '(AAAAG)', '(CCTG)(TCTG)']


def gen_similarity(ns, outdir):
    """summaries/state_similarity.csv and the high-similarity warnings, produced by the reference's own
    CallerWrapper.check_high_similarity (src/caller/wrapper.py:122-160; PoreModel.get_diffs_for_all,
    src/squiggler/pore_model.py:49-71) -- called unbound on a stand-in object that only carries locus.path."""
    import contextlib
    import io
    import tempfile
    import types
    W = ns.wrapper.CallerWrapper
    out = {}
    for seq in SIMILARITY_SEQUENCES:
        tmp = tempfile.mkdtemp(prefix='warpstr_sim_')
        os.makedirs(os.path.join(tmp, ns.templates.SUMMARY_SUBDIR))
        fake = types.SimpleNamespace(locus=types.SimpleNamespace(path=tmp), reverse_uniq_sequence=W.reverse_uniq_sequence)
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf):
                tp, rp = W.check_high_similarity(fake, seq)
        except Exception as e:  # noqa: BLE001 -- e.g. one-base units: the k-mer list runs past the repeated pattern
            out[seq] = dict(error=type(e).__name__)
            continue
        with open(os.path.join(tmp, ns.templates.SUMMARY_SUBDIR, 'state_similarity.csv')) as f:
            csv_text = f.read()
        out[seq] = dict(csv=csv_text, stdout=buf.getvalue(),
                        template_problems=[dict(pattern=p['pattern'], mean_diff=float(p['mean_diff']), median_diff=float(p['median_diff'])) for p in tp],
                        reverse_problems=[dict(pattern=p['pattern'], mean_diff=float(p['mean_diff']), median_diff=float(p['median_diff'])) for p in rp])
    with open(os.path.join(outdir, 'similarity.json'), 'w') as f:
        json.dump(dict(min_state_similarity=float(ns.wrapper.caller_config.min_state_similarity), summary_subdir=ns.templates.SUMMARY_SUBDIR,
                       cases=out), f, indent=1)


def _genotype_length_sets():
    """Seeded per-read allele-length sets for the step-4 fixtures: (name, numpy seed for the mixture, values)."""
    rng = np.random.default_rng(77)
    two = lambda a, na, b, nb, sd: [int(v) for v in np.concatenate([np.round(rng.normal(a, sd, na)), np.round(rng.normal(b, sd, nb))])]
    sets = [
        ('two_far_alleles', 1, two(30, 22, 50, 19, 0.8)),
        ('two_close_alleles', 2, two(44, 30, 40, 26, 0.7)),
        ('homozygous_noisy', 3, [int(v) for v in np.round(rng.normal(60, 1.0, 48))]),
        ('single_value', 4, [40] * 12),
        ('five_values_no_filter', 5, [21, 22, 21, 90, 22]),
        ('six_values', 6, [21, 22, 21, 35, 22, 36]),
        ('outliers_filtered', 7, two(35, 25, 47, 25, 0.9) + [200, 2]),
        ('minor_component_below_min_weight', 8, two(30, 46, 50, 4, 0.6)),
        ('two_values_only', 9, [17, 23]),
        ('shuffled_two_alleles', 10, None),
    ]
    v = two(25, 18, 33, 21, 1.1)
    rng.shuffle(v)
    sets[-1] = ('shuffled_two_alleles', 10, [int(x) for x in v])
    return sets


def _complex_tables():
    """Per-read repeat-unit count tables (the columns store_collapsed writes, src/caller/overview.py:11-34)."""
    rng = np.random.default_rng(78)
    def table(n1, c1, n2, c2, sd, names, extra=()):
        rows = [np.round(rng.normal(c1, sd)).astype(int) for _ in range(n1)] + [np.round(rng.normal(c2, sd)).astype(int) for _ in range(n2)]
        rows += [np.array(e) for e in extra]
        order = rng.permutation(len(rows))
        d = {name: [int(rows[i][k]) for i in order] for k, name in enumerate(names)}
        d['reverse'] = [bool(rng.integers(0, 2)) for _ in order]
        return d
    return [
        ('two_alleles_two_units', 11, table(20, (12, 30), 18, (20, 22), 0.7, ['AGC', 'CGC'])),
        ('two_alleles_three_units', 12, table(16, (10, 5, 14), 16, (18, 5, 9), 0.8, ['main_CAGG', 'CAGA', 'CA'])),
        ('homozygous_two_units', 13, table(40, (15, 8), 0, (0, 0), 0.8, ['AAGGG', 'AAAGG'])),
        ('five_rows', 14, table(3, (7, 9), 2, (12, 4), 0.5, ['CCTG', 'TCTG'])),
        ('outlier_rows_dropped_two_alleles', 15, table(20, (12, 30), 20, (22, 18), 0.7, ['AGC', 'CGC'], extra=[(90, 30), (12, 2)])),
        ('outlier_rows_dropped_homozygous', 16, table(30, (15, 8), 0, (0, 0), 0.7, ['AAGGG', 'AAAGG'], extra=[(60, 8), (15, 40)])),
        ('one_unit_only', 17, {'AGC': [5, 6, 7, 5, 6, 7, 8], 'reverse': [False] * 7}),
        ('homozygous_nothing_dropped', 18, {'AGC': [15, 16] * 9, 'CGC': [8, 8, 9] * 6, 'reverse': [False, True] * 9}),
        ('homozygous_identical_rows', 19, {'AGC': [15] * 10, 'CGC': [8] * 10, 'reverse': [True] * 10}),
        ('homozygous_early_row_dropped', 20, {'AGC': [40] + [15, 16] * 10, 'CGC': [8] + [8, 9] * 10, 'reverse': [False] * 21}),
    ]


def gen_genotype(outdir):
    """Step 4 (src/genotyper/genotyping.py): the reference's own run_genotyping / run_genotyping_overview (incl. the basecall
    alleles of load_predictions 95-103) / run_genotyping_complex on seeded inputs.  scikit-learn's mixture draws from numpy's
    global generator, so each case seeds it (np.random.seed) right before the call; the plots are switched off."""
    import contextlib
    import io
    import tempfile
    import pandas as pd
    from _ref_import import REFERENCE_ROOT
    old = os.getcwd()
    os.chdir(REFERENCE_ROOT)
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        from src.genotyper import genotyping
    finally:
        os.chdir(old)
        sys.path.remove(REFERENCE_ROOT)
    genotyping.genotyping_config.visualize = False
    genotyping.plot_complex_repeats = lambda *a, **k: None
    out = dict(min_weight=float(genotyping.genotyping_config.min_weight), std_filter=float(genotyping.genotyping_config.std_filter),
               simple=[], overview=[], complex=[])
    for name, seed, vals in _genotype_length_sets():
        np.random.seed(seed)
        gt = genotyping.run_genotyping(list(vals))
        out['simple'].append(dict(name=name, seed=seed, values=vals, group1=[int(v) for v in gt.group1], group2=[int(v) for v in gt.group2],
                                  predictions=[int(v) for v in gt.predictions], alleles=[gt.first_allele, gt.second_allele],
                                  sizes=[gt.first_allele_sz, gt.second_allele_sz]))
    # overview tables: with and without the basecall columns; rows that were not `saved` are skipped
    sets = {n: v for n, _, v in _genotype_length_sets()}
    rng = np.random.default_rng(79)
    for name, seed, vals, with_bc in (('warpstr_only', 21, sets['two_far_alleles'], False), ('with_basecalls', 22, sets['two_close_alleles'], True),
                                      ('with_basecalls_homozygous', 23, sets['homozygous_noisy'], True)):
        n = len(vals)
        saved = [bool(rng.random() < 0.85) for _ in range(n)]
        cols = dict(read_name=[f'read{i:03d}' for i in range(n)], saved=saved, results=[v if s else -1 for v, s in zip(vals, saved)])
        if with_bc:
            l_end = [int(rng.integers(200, 400)) for _ in range(n)]
            cols['l_seq_end'] = l_end
            cols['r_seq_start'] = [le + 3 * v + int(rng.integers(-4, 5)) for le, v in zip(l_end, vals)]
        df = pd.DataFrame(cols).set_index('read_name')
        tmp = tempfile.mkdtemp(prefix='warpstr_gt_')
        os.makedirs(os.path.join(tmp, 'predictions'))
        buf = io.StringIO()
        np.random.seed(seed)
        with contextlib.redirect_stdout(buf):
            genotyping.run_genotyping_overview(df, tmp, None)
        with open(os.path.join(tmp, 'predictions', 'alleles.csv')) as f:
            text = f.read()
        out['overview'].append(dict(name=name, seed=seed, columns={k: v for k, v in cols.items()}, alleles_csv=text, stdout=buf.getvalue()))
    for name, seed, table in _complex_tables():
        df = pd.DataFrame.from_dict(table)
        tmp = tempfile.mkdtemp(prefix='warpstr_gtc_')
        os.makedirs(os.path.join(tmp, 'predictions', 'complexSTR_analysis'))
        os.makedirs(os.path.join(tmp, 'summaries'))
        buf = io.StringIO()
        np.random.seed(seed)
        rec = dict(name=name, seed=seed, table=table)
        try:
            with contextlib.redirect_stdout(buf):
                genotyping.run_genotyping_complex(tmp, df)
        except Exception as e:  # noqa: BLE001 -- e.g. the label lookup of find_nearest on a table that lost rows
            rec['error'] = type(e).__name__
        path = os.path.join(tmp, 'predictions', 'complexSTR_analysis', 'complex_alleles.csv')
        rec['complex_alleles_csv'] = open(path).read() if os.path.exists(path) else None
        rec['stdout'] = buf.getvalue()
        out['complex'].append(rec)
    with open(os.path.join(outdir, 'genotype.json'), 'w') as f:
        json.dump(out, f, indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--alt', action='store_true')
    ap.add_argument('--only', default=None)
    args = ap.parse_args()
    if args.alt:
        for case in ALT_CASES:
            if args.only and case[0] != args.only:
                continue
            # one config per process: the reference's config is an import-time singleton
            if args.only is None:
                os.system(f'{sys.executable} {os.path.abspath(__file__)} --alt --only {case[0]}')
                continue
            name, pattern, fl, fseed, T, n, rseed, lohi, cfg = case
            ns = import_reference(**cfg)
            print(name, cfg)
            gen_case(ns, name, pattern, fl, fseed, T, n, rseed, lohi, HERE)
            with open(os.path.join(HERE, f'{name}.config.json'), 'w') as f:
                json.dump(cfg, f)
        return
    ns = import_reference()
    for case in CASES:
        if args.only and case[0] != args.only:
            continue
        print(case[0])
        gen_case(ns, *case, HERE)
    if not args.only or args.only == 'real_aaat':
        print('real_aaat')
        gen_real(ns, HERE)
    if not args.only or args.only == 'similarity':
        print('similarity')
        gen_similarity(ns, HERE)
    if not args.only or args.only == 'genotype':
        print('genotype')
        gen_genotype(HERE)
    if not args.only:
        gen_units(ns, HERE)
        gen_negative(ns, HERE)


if __name__ == '__main__':
    main()
