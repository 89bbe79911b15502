"""Host-side logic against the fixtures recorded from the upstream code.  CPU only."""
import json
import os

import numpy as np
import pytest

from tests.helpers import DEFAULT_CASES, GOLDEN, load_case
from warpstr_amd.automata import compile_automaton, reverse_pattern
from warpstr_amd.caller import sequence_from_trace
from warpstr_amd.pore_model import PoreModel, default_pore_model
from warpstr_amd.signal_prep import brute_remove, process_raw
from warpstr_amd.units import break_into_units, collapse_repeats


@pytest.mark.parametrize('case', DEFAULT_CASES)
def test_automaton_tables_match_reference(case):
    z = load_case(case)
    fl = [str(s) for s in z['flanks']]
    pat = str(z['pattern'])
    for tag, seq in (('t', fl[0] + pat + fl[1]), ('r', fl[2] + reverse_pattern(pat) + fl[3])):
        a = compile_automaton(seq)
        assert a.n_states == len(z[f'{tag}_value']) and a.endstate == int(z[f'{tag}_endstate'])
        assert np.array_equal(a.value, z[f'{tag}_value'])       # bit-identical levels
        assert np.array_equal(a.seq_idx, z[f'{tag}_seq_idx'])
        assert np.array_equal(a.pred_ptr, z[f'{tag}_pred_ptr'])
        assert np.array_equal(a.pred_idx, z[f'{tag}_pred_idx'])
        assert np.array_equal(a.repeat_mask, z[f'{tag}_mask'])
        assert a.kmers == [str(k) for k in z[f'{tag}_kmers']]


@pytest.mark.parametrize('case', DEFAULT_CASES)
def test_sequence_from_trace(case):
    z = load_case(case)
    fl = [str(s) for s in z['flanks']]
    pat = str(z['pattern'])
    tabs = [compile_automaton(fl[0] + pat + fl[1]), compile_automaton(fl[2] + reverse_pattern(pat) + fl[3])]
    for i in range(int(z['n_reads'])):
        rev = int(z['reverse'][i])
        seq, rseq = [str(s) for s in z[f'r{i}_seq']]
        assert sequence_from_trace(tabs[rev], int(z['flank_length']), z[f'r{i}_trace1'], bool(rev)) == seq
        assert sequence_from_trace(tabs[rev], int(z['flank_length']), z[f'r{i}_trace2'], bool(rev)) == rseq


def test_units_and_collapse():
    with open(os.path.join(GOLDEN, 'units.json')) as f:
        ref = json.load(f)
    for pat, exp in ref.items():
        assert reverse_pattern(pat) == exp['reverse']
        units, repeat_units, offsets = break_into_units(pat)
        assert (units, repeat_units, offsets) == (exp['units'], exp['repeat_units'], exp['offsets'])
        for seq, counts in exp['collapse']:
            assert collapse_repeats(seq, repeat_units, offsets) == counts
    # a top-level optional block yields an empty repeat unit; upstream spins forever, we refuse
    _, ru, off = break_into_units('(ARC)TT{GG}A(CCG)')
    with pytest.raises(RuntimeError):
        collapse_repeats('AACGGA', ru, off, max_iter=1000)


def test_signal_prep_and_pore_model():
    z = np.load(os.path.join(GOLDEN, 'signal_prep.npz'))
    cleaned = brute_remove(z['raw'])
    assert cleaned.dtype == z['raw'].dtype and np.array_equal(cleaned, z['cleaned'])
    assert np.array_equal(process_raw(z['raw']), z['norm'])
    assert np.array_equal(process_raw(z['raw'], (100, 220)), z['norm'][100:221])
    assert np.array_equal(default_pore_model().level_norm, z['pore_level_norm'])


def test_pore_model_tsv_roundtrip(tmp_path):
    pm = default_pore_model()
    raw = np.load(os.path.join(os.path.dirname(GOLDEN), '..', 'warpstr_amd', 'data', 'r94_6mer_level_mean.npy'))
    import itertools
    p = tmp_path / 'model.tsv'
    with open(p, 'w') as f:
        f.write('kmer\tlevel_mean\tlevel_stdv\n')
        for k, v in zip(itertools.product('ACGT', repeat=6), raw):
            f.write(''.join(k) + f'\t{float(v)!r}\t1.0\n')
    pm2 = PoreModel(str(p))
    assert np.array_equal(pm2.level_norm, pm.level_norm)
    assert pm2.get_value('ACGTAC') == pm.get_value('ACGTAC')


def test_config_to_loci_like_upstreams_loop(tmp_path):
    """A WarpSTR configuration with upstream's keys (test/config_template.yaml, example/config.yaml) -> the locus objects the
    several-loci driver takes: <output>/<name> directories (src/schemas/locus.py:18-46), per-locus flank_length falling back to
    the global one (WarpSTR.py:33-37 passes every locus dict to Locus)."""
    from warpstr_amd.config import load_config
    from warpstr_amd.wrapper import loci_from_config
    p = tmp_path / 'cfg.yaml'
    p.write_text('reference_path: /nowhere/GRCh38.fa\noutput: %s\ninputs:\n  - path: test/test_input\n    runs: test_run1\n'
                 'tr_region_calling: True\ngenotyping: True\nthreads: 3\nrescaling:\n  threshold: 0.4\n'
                 'guppy_config:\n  path: /x\nloci:\n  - name: Human_STR_1108232\n    coord: chr4:183178378-183178421\n'
                 '    sequence: (aaat)\n  - name: DM2\n    coord: chr3:129,172,577-129,172,732\n'
                 '    sequence: ((CAGG){CAGM})(CAGA)(CA)\n    flank_length: 90\n' % (tmp_path / 'out'))
    cfg = load_config(str(p))
    loci = loci_from_config(cfg)
    assert [(l.name, l.sequence, l.flank_length) for l in loci] == [('Human_STR_1108232', '(AAAT)', 110), ('DM2', '((CAGG){CAGM})(CAGA)(CA)', 90)]
    assert loci[1].path == str(tmp_path / 'out' / 'DM2')
    assert cfg.threads == 3 and cfg.rescaler.threshold == 0.4 and cfg.rescaler.max_std == 0.5 and cfg.raw['genotyping'] is True
    (tmp_path / 'nomotif.yaml').write_text('output: x\nloci:\n  - name: HD\n    motif: AGC,CGC\n')
    import pytest
    with pytest.raises(ValueError, match='explicit `sequence`'):
        load_config(str(tmp_path / 'nomotif.yaml'))


def test_config_defaults(tmp_path):
    from warpstr_amd.config import load_config
    p = tmp_path / 'cfg.yaml'
    p.write_text('output: out\nthreads: 4\nrescaling:\n  threshold: 0.4\nloci:\n  - name: L1\n    sequence: (agc)\n'
                 '    coord: chr1:1-2\n  - name: L2\n    sequence: (AAAT)\n    flank_length: 40\n')
    c = load_config(str(p))
    assert c.caller.min_values_per_state == 4 and c.caller.states_in_segment == 6
    assert c.rescaler.threshold == 0.4 and c.rescaler.max_std == 0.5 and c.rescaler.method == 'mean'
    assert [(l.name, l.sequence, l.flank_length) for l in c.loci] == [('L1', '(AGC)', 110), ('L2', '(AAAT)', 40)]


def test_overview_outputs(tmp_path):
    import pandas as pd

    from warpstr_amd import overview as ov
    loc = tmp_path / 'L'
    (loc / 'expected_signals').mkdir(parents=True)
    (loc / 'expected_signals' / 'sequences.csv').write_text(
        'type,sequence\nleft_flank_template,AAAC\nright_flank_template,CCCG\nleft_flank_reverse,CGGG\n'
        'right_flank_reverse,GTTT\ntemp_ref_pattern,AGCAGC\nrev_ref_pattern,GCTGCT\n')
    pd.DataFrame({'read_name': ['a', 'b', 'c'], 'run_id': [0, 0, 0], 'reverse': [False, True, False],
                  'saved': [1, 0, 1], 'l_start_raw': [0, 0, 0], 'r_end_raw': [9, 9, 9],
                  'results_old': [5, 5, 5]}).to_csv(loc / 'overview.csv', index=False)
    assert ov.load_flanks(str(loc)) == ('AAAC', 'CCCG', 'CGGG', 'GTTT')
    path, df = ov.load_overview(str(loc))
    df = ov.store_results(path, df, [('AGC', 'AGCAGC'), ('AG', 'AGCA')], [(0.1, 0.2), (0.3, 0.4)], str(loc))
    out = pd.read_csv(path)
    assert list(out['results']) == [6, -1, 4] and list(out['orig']) == [3, -1, 2]
    assert list(out['dtw_cost2']) == [0.2, -1.0, 0.4] and 'results_old' not in out.columns
    allf = (loc / 'predictions' / 'sequences' / 'all.fasta').read_text()
    assert allf == '>a\nAGCAGC\n\n>c\nAGCA\n\n'
    assert (loc / 'predictions' / 'sequences' / 'sequences_reverse.fasta').read_text() == ''


def test_genotyper_two_alleles_and_homozygous(tmp_path):
    from warpstr_amd import _lib
    from warpstr_amd.genotyper import call_alleles, genotype_results, trim_outliers, write_alleles_csv
    rng = np.random.default_rng(3)
    vals = list(np.round(rng.normal(44, 1.0, 40)).astype(int)) + list(np.round(rng.normal(120, 1.5, 35)).astype(int))
    gt = call_alleles(vals, random_state=0)
    assert gt.heterozygous and sorted(abs(a - b) <= 2 for a, b in zip(sorted(gt.alleles), (44, 120))) == [True, True]
    assert gt.support(0) + gt.support(1) == len(trim_outliers(vals, 2))
    hom = call_alleles([40] * 12, random_state=0)
    assert not hom.heterozygous and hom.alleles == (40, '-')
    near = call_alleles(list(np.round(rng.normal(60, 0.8, 50)).astype(int)) + [200], random_state=0)
    assert not near.heterozygous and abs(near.allele(0) - 60) <= 1          # the outlier is set aside (+-2 sigma)
    rec = np.zeros(6, dtype=_lib.RESULT_DTYPE)
    rec['len2'] = [30, 30, 30, 30, 99, 30]
    rec['status'] = [0, 0, 0, 0, 7, 0]
    assert genotype_results(rec).alleles == (30, '-')
    p = write_alleles_csv(str(tmp_path), gt)
    assert open(p).read().splitlines()[0].startswith('WarpSTR_allele1,')


def test_state_similarity_matches_reference(tmp_path, capsys):
    """summaries/state_similarity.csv, the warnings and the returned problem lists against what the reference's
    CallerWrapper.check_high_similarity produced (tests/golden/similarity.json, generate_golden.py: gen_similarity)."""
    import json
    import os

    import pytest

    from warpstr_amd.caller import CallerConfig, CallerWrapper
    from warpstr_amd.pore_model import default_pore_model
    with open(os.path.join(GOLDEN, 'similarity.json')) as f:
        fx = json.load(f)
    assert CallerConfig().min_state_similarity == fx['min_state_similarity']
    cw = CallerWrapper.__new__(CallerWrapper)  # no GPU: only the host-side check
    cw.caller_config, cw.pore_model, cw.locus = CallerConfig(), default_pore_model(), None
    for seq, want in fx['cases'].items():
        out_dir = tmp_path / str(abs(hash(seq)))
        if 'error' in want:
            with pytest.raises(IndexError):
                cw.check_high_similarity(seq, str(out_dir))
            continue
        capsys.readouterr()
        tp, rp = cw.check_high_similarity(seq, str(out_dir))
        assert capsys.readouterr().out == want['stdout']
        assert (out_dir / 'state_similarity.csv').read_text() == want['csv']
        for got, ref in ((tp, want['template_problems']), (rp, want['reverse_problems'])):
            assert [p['pattern'] for p in got] == [p['pattern'] for p in ref]
            for g, r in zip(got, ref):
                assert abs(g['mean_diff'] - r['mean_diff']) < 1e-12 and abs(g['median_diff'] - r['median_diff']) < 1e-12


def test_caller_results_are_lazy_and_list_like():
    """CallerWrapper.run's return value: CallerResult objects (src/caller/caller.py:46-51) built on access from the batch's
    records and ASCII buffers; failed reads raise (the reference loses the whole batch) or give NaN records."""
    import pytest

    from warpstr_amd import _lib
    from warpstr_amd.caller import CallerResult, CallerResults, ReadCallError
    rec = np.zeros(3, dtype=_lib.RESULT_DTYPE)
    rec['len1'], rec['len2'] = [3, 2, 0], [4, 1, 0]
    rec['cost1'], rec['cost2'] = [0.5, 0.25, np.nan], [0.4, 0.2, np.nan]
    rec['status'] = [0, 0, 3]
    seq1 = b'ACGxxxxxTTyyyyyyzzzz'
    seq2 = b'ACGTxxxxGyyyyyyyzzzz'
    offsets = np.array([0, 8, 16])
    res = CallerResults(['a', 'b', 'c'], rec, offsets, seq1, seq2, 'nan')
    assert len(res) == 3 and res[0] == CallerResult('ACG', 0.5, 'ACGT', 0.4) and res[1] == CallerResult('TT', 0.25, 'G', 0.2)
    assert res[-1].seq == '' and np.isnan(res[2].resc_cost)
    assert [r.resc_seq for r in res] == ['ACGT', 'G', ''] and [r.seq for r in res[0:2]] == ['ACG', 'TT']
    assert res.lengths()[1].tolist() == [4, 1, 0]
    strict = CallerResults(['a', 'b', 'c'], rec, offsets, seq1, seq2, 'raise')
    with pytest.raises(ReadCallError):
        strict.check()
    with pytest.raises(ReadCallError):
        strict[2]
    assert strict[0].seq == 'ACG'


def test_seam_helper_loops():
    """csrc/seam_helper.c: pointers / lengths / strand flags of a workload of ReadSignal objects, refusal of anything that
    is not a contiguous float64 1-d buffer (the caller then converts in Python), and the packing of called sequences."""
    from warpstr_amd import _lib, build, caller
    from warpstr_amd.caller import ReadSignal
    if not os.path.exists(build.SEAM_LIB):  # (a clean clone: the helper is a few seconds of gcc, the HIP library minutes)
        import subprocess
        import sysconfig
        subprocess.check_call(['gcc', '-O2', '-shared', '-fPIC', '-I', sysconfig.get_paths()['include'], build.SEAM_SRC, '-o', build.SEAM_LIB])
        caller._SEAM = False
    seam = caller._seam()
    assert seam is not None, 'python -m warpstr_amd.build builds warpstr_amd/_seam_helper.so'
    rng = np.random.default_rng(1)
    work = [ReadSignal(f'r{i}', bool(i % 3 == 0), rng.standard_normal(5 + i)) for i in range(40)]
    ptrs, lens, aut = np.empty(40, np.uintp), np.empty(40, np.int64), np.empty(40, np.int32)
    assert seam.wsx_seam_collect(work, b'signal', b'reverse', _lib.ptr(ptrs), _lib.ptr(lens), _lib.ptr(aut), None) == 40
    assert all(int(ptrs[i]) == work[i].signal.ctypes.data and lens[i] == 5 + i and aut[i] == (i % 3 == 0) for i in range(40))
    assert seam.wsx_seam_collect(tuple(work), b'signal', b'reverse', _lib.ptr(ptrs), _lib.ptr(lens), _lib.ptr(aut), None) == 40
    for bad in ([1.0, 2.0], np.zeros(4, np.float32), np.zeros(8)[::2], np.zeros((2, 2)), np.zeros(3, np.int64)):
        w2 = list(work)
        w2[9] = ReadSignal('x', False, bad)
        assert seam.wsx_seam_collect(w2, b'signal', b'reverse', _lib.ptr(ptrs), _lib.ptr(lens), _lib.ptr(aut), None) == -10
    assert seam.wsx_seam_collect([], b'signal', b'reverse', None, None, None, None) == 0
    class Fresh:  # a workload whose `signal` is built per access: the keep-alive list holds what the pointers refer to
        reverse = False

        @property
        def signal(self):
            return np.full(7, 3.5)
    keep = []
    assert seam.wsx_seam_collect([Fresh(), Fresh()], b'signal', b'reverse', _lib.ptr(ptrs), _lib.ptr(lens), _lib.ptr(aut), keep) == 2
    assert len(keep) == 2 and all(int(ptrs[i]) == keep[i].ctypes.data and lens[i] == 7 for i in range(2))
    src = np.frombuffer(b'AAAACCCCGGGGTTTT', np.uint8).copy()
    off = np.array([0, 4, 8, 12, 16], np.int64)
    ln = np.array([2, 0, 4, 1], np.int32)
    pos, out = np.empty(5, np.int64), np.empty(7, np.uint8)
    assert seam.wsx_seam_pack_sequences(_lib.ptr(src), _lib.ptr(off), _lib.ptr(ln), 4, 4, _lib.ptr(out), _lib.ptr(pos)) == 7
    assert bytes(out) == b'AAGGGGT' and pos.tolist() == [0, 2, 2, 6, 7]
