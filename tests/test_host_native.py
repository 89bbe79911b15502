"""The native host library (warpstr_amd/csrc/host_loci.cpp -> _host_loci.so) against the Python forms it restates: automaton
tables field for field, repr(float), overview.csv / FASTA / complex-unit files byte for byte with the pandas path -- and that it
declines what it cannot be sure about."""
import filecmp
import os
import shutil

import numpy as np
import pandas as pd
import pytest

from warpstr_amd import _hostlib, automata, overview as ov
from warpstr_amd.pore_model import default_pore_model

pytestmark = pytest.mark.skipif(_hostlib.lib() is None, reason='warpstr_amd/_host_loci.so is not built')

UNITS = ['AGC', 'AAAT', 'GGCCCC', 'CAG', 'CTG', 'CCTG', 'NGC', 'RY', 'CAGM', 'AAGGG', 'GAA', 'TTTTA', 'GCN', 'CGG', 'AT', 'ATTCT', 'A',
         'BD', 'HV', 'SWK']


def random_pattern(rng):
    pat = ''
    for _ in range(int(rng.integers(1, 4))):
        unit = UNITS[int(rng.integers(len(UNITS)))]
        r = rng.random()
        if r < 0.2:
            unit = '(' + unit + '){' + UNITS[int(rng.integers(len(UNITS)))] + '}'
        elif r < 0.3:
            unit = unit + '{' + UNITS[int(rng.integers(len(UNITS)))] + '}'
        elif r < 0.35:
            unit = '(' + unit + ')' + UNITS[int(rng.integers(len(UNITS)))]
        pat += '(' + unit + ')'
        if rng.random() < 0.4:
            pat += ''.join('ACGT'[i] for i in rng.integers(0, 4, size=int(rng.integers(1, 14))))
    return pat


def same_table(a, b):
    assert (a.n_states, a.endstate, a.repstart, a.repend) == (b.n_states, b.endstate, b.repstart, b.repend)
    for f in ('value', 'seq_idx', 'pred_ptr', 'pred_idx', 'repeat_mask', 'last_base'):
        x, y = getattr(a, f), getattr(b, f)
        assert x.dtype == y.dtype and np.array_equal(x, y), f
    assert a.kmers == b.kmers
    assert [sorted(s) for s in a.succ] == [sorted(s) for s in b.succ]


def test_automata_equal_the_python_compiler_on_random_loci():
    rng = np.random.default_rng(5)
    pm = default_pore_model()
    n = 0
    for _ in range(600):
        pat = random_pattern(rng)
        fl = int(rng.integers(6, 130))
        left = ''.join('ACGT'[i] for i in rng.integers(0, 4, size=fl))
        right = ''.join('ACGT'[i] for i in rng.integers(0, 4, size=fl))
        for full in (left + pat + right, left + automata.reverse_pattern(pat) + right):
            try:
                want = automata.compile_automaton(full, pm, native=False)
            except Exception:  # noqa: BLE001 -- a pattern the compiler refuses: the library must decline it too
                assert _hostlib.compile_automaton(full, pm) is None
                continue
            got = _hostlib.compile_automaton(full, pm)
            assert got is not None, full
            same_table(got, want)
            n += 1
    assert n > 1000


@pytest.mark.parametrize('pattern', ['', 'ACG', 'ACGTAC)AGC(ACGTAC', 'ACGTACGT}AGC{ACGTAC', 'ACGTACxGTACGTAC', 'acgtacgtacgt', 'ACGTAC(AGC'])
def test_patterns_the_python_compiler_raises_on_are_declined_or_equal(pattern):
    pm = default_pore_model()
    try:
        want = automata.compile_automaton(pattern, pm, native=False)
    except Exception:  # noqa: BLE001
        assert _hostlib.compile_automaton(pattern, pm) is None
        with pytest.raises(Exception):
            automata.compile_automaton(pattern, pm)  # (the default path ends in the Python compiler's own error)
        return
    same_table(_hostlib.compile_automaton(pattern, pm), want)


def test_float_text_is_pythons_repr():
    rng = np.random.default_rng(1)
    xs = [0.0, -0.0, 1.0, -1.0, 0.1, 1e-4, 9.999e-5, 1e-5, 1.5e-7, 1e15, 1e16, 123456789012345680.0, 1e22, 5e-324, 1.7976931348623157e308,
          2.0 ** -30, 2.0 ** 70, 0.30000000000000004, 100.0, 1234.5, float('inf'), float('-inf')]
    xs += list(rng.standard_normal(2000)) + list(np.exp(rng.uniform(-40, 40, 2000))) + list(rng.integers(-10 ** 6, 10 ** 6, 500) / 8.0)
    xs += [float(np.float64(v)) for v in np.round(rng.uniform(0, 5, 500), 3)]
    for x in xs:
        assert _hostlib.format_float(x) == repr(float(x)), x
    col = pd.DataFrame({'x': np.array(xs[:200])}).to_csv(index=False).split('\n')[1:-1]
    assert col == [_hostlib.format_float(x) for x in xs[:200]]


def _random_overview(rng, n, kind):
    """A table as the earlier pipeline steps leave it (pandas-written), with the column kinds real runs have."""
    names = [f'{rng.integers(0, 16**8):08x}-{rng.integers(0, 16**4):04x}-read{i}' for i in range(n)]
    df = pd.DataFrame({'read_name': names, 'run_id': rng.choice(['run_0', 'runB'], n) if kind % 2 else rng.integers(0, 3, n),
                       'reverse': rng.random(n) < 0.5, 'sam_dist': rng.integers(-50, 50, n),
                       'saved': (rng.random(n) < 0.8) if kind % 3 else (rng.random(n) < 0.8).astype(int)})
    lo = rng.integers(0, 50000, n)
    df['l_start_raw'], df['r_end_raw'] = lo, lo + rng.integers(200, 4000, n)
    if kind >= 2:   # floats with blanks on unsaved rows, a string column with blanks, scores
        df['l_start_raw'] = np.where(df['saved'].astype(bool), df['l_start_raw'], np.nan)
        df['r_end_raw'] = np.where(df['saved'].astype(bool), df['r_end_raw'], np.nan)
        df['score'] = rng.standard_normal(n) * 10.0 ** rng.integers(-6, 6, n)
        df['note'] = np.where(rng.random(n) < 0.5, 'ok', None)
        df['flag'] = np.where(rng.random(n) < 0.3, None, rng.random(n) < 0.5)
    if kind >= 4:   # a re-run: the result columns are there already, and a stale one
        df['results'], df['orig'], df['dtw_cost1'], df['dtw_cost2'], df['result_old'] = 7, 8, 0.5, 0.25, 'x'
        df['after'] = rng.integers(0, 9, n)
    return df


@pytest.mark.parametrize('kind', range(6))
def test_overview_in_and_out_equals_the_pandas_path(tmp_path, kind):
    rng = np.random.default_rng(100 + kind)
    for trial in range(8):
        n = int(rng.integers(1, 60))
        df = _random_overview(rng, n, kind)
        if not df['saved'].astype(bool).any():
            df.loc[0, 'saved'] = True if df['saved'].dtype == bool else 1
            if kind >= 2:
                df.loc[0, ['l_start_raw', 'r_end_raw']] = [5.0, 900.0]
        a, b = str(tmp_path / f'a{trial}'), str(tmp_path / f'b{trial}')
        for d in (a, b):
            os.makedirs(d)
            df.to_csv(os.path.join(d, 'overview.csv'), index=False)
        nat = _hostlib.NativeOverview.open(os.path.join(a, 'overview.csv'))
        assert nat is not None, _hostlib.NativeOverview.last_refusal
        path, ref = ov.load_overview(b)
        saved = np.flatnonzero(np.asarray(ref['saved']).astype(bool))
        assert nat.saved.tolist() == saved.tolist()
        assert nat.names == [str(x) for x in ref.index.to_numpy()[saved]]
        assert nat.reverse.tolist() == np.asarray(ref['reverse'])[saved].astype(bool).tolist()
        assert nat.lo.tolist() == np.asarray(ref['l_start_raw'])[saved].astype(np.int64).tolist()
        assert nat.hi.tolist() == np.asarray(ref['r_end_raw'])[saved].astype(np.int64).tolist()
        assert nat.run_id == [str(x) for x in np.asarray(ref['run_id'])[saved]]
        ns = len(saved)
        seqs = [(''.join(rng.choice(list('ACGT'), int(rng.integers(0, 40)))), ''.join(rng.choice(list('ACGT'), int(rng.integers(0, 40)))))
                for _ in range(ns)]
        costs = [(float(rng.random() * 3), float(np.exp(rng.uniform(-12, 3)))) for _ in range(ns)]
        want = ov.store_results(path, ref, seqs, costs, b, write=True)
        len1, len2 = [len(s[0]) for s in seqs], [len(s[1]) for s in seqs]
        seq2 = np.frombuffer(''.join(s[1] for s in seqs).encode(), np.uint8)
        off2 = np.cumsum([0] + len2)[:-1]
        text = nat.store(a, len1, len2, [c[0] for c in costs], [c[1] for c in costs], seq2, off2, write=True)
        for rel in ('overview.csv', 'predictions/sequences/all.fasta', 'predictions/sequences/sequences_template.fasta',
                    'predictions/sequences/sequences_reverse.fasta'):
            assert filecmp.cmp(os.path.join(a, rel), os.path.join(b, rel), shallow=False), (kind, trial, rel)
        assert text == open(os.path.join(b, 'overview.csv')).read()
        got = ov.table_from_text(text)
        pd.testing.assert_frame_equal(got, want)
        nat.close()


@pytest.mark.parametrize('edit, why', [
    (lambda t: t.replace('\n', '\r\n'), 'carriage'),
    (lambda t: t.replace(',3,', ',03,', 1) if ',3,' in t else t.replace(',1,', ',01,', 1), 're-formatted'),
    (lambda t: t.replace('run_0', '"run,0"', 1), 'quoted'),
    (lambda t: t.replace('run_0', 'NA', 1), 're-formatted'),
    (lambda t: t.replace('True', 'true', 1), 're-formatted'),
    (lambda t: t.replace('0.5', '+0.5', 1), 're-formatted'),
    (lambda t: t.replace('0.5', '.5', 1), 're-formatted'),
    (lambda t: t.replace('0.5', '1e400', 1), 're-formatted'),
    (lambda t: t[:-1], 'newline'),
    (lambda t: t + '\n', 'blank'),
    (lambda t: t.replace('read_name', 'read_name,read_name', 1), ''),
    (lambda t: t.replace('saved', 'kept', 1), 'missing'),
])
def test_tables_pandas_would_change_are_declined(tmp_path, edit, why):
    df = pd.DataFrame({'read_name': ['r1', 'r2', 'r3'], 'run_id': ['run_0', 'run_0', 'run_1'], 'reverse': [True, False, True],
                       'saved': [1, 1, 0], 'l_start_raw': [3, 13, 1], 'r_end_raw': [900, 950, 1000], 'score': [0.5, 1.25, 3.0]})
    text = edit(df.to_csv(index=False))
    p = tmp_path / 'overview.csv'
    p.write_bytes(text.encode())
    assert _hostlib.NativeOverview.open(str(p)) is None
    assert why in _hostlib.NativeOverview.last_refusal


def test_missing_overview_raises_upstreams_error(tmp_path):
    with pytest.raises(FileNotFoundError, match='Not found the overview file'):
        _hostlib.NativeOverview.open(str(tmp_path / 'overview.csv'))


def test_float_columns_come_back_as_pandas_converter_leaves_them(tmp_path):
    """read_csv's default float converter is not correctly rounded and to_csv writes what it made of a cell: a float column
    changes on its way through pandas.  The writer does to every float cell what pandas does (pandas_strtod in host_loci.cpp)."""
    rng = np.random.default_rng(3)
    cells = ['0.50', '1e5', '5.', '3', '0.0015732835352270625', '123456789.123456789123', '1E-7', '2.5e+300', '1e-320', '4.9e-324',
             '0.1', '-0.0', '17', '1.7976931348623157e308', '0.30000000000000004', '9007199254740993.0', '1e22', '1e23', '8.5e-5', 'inf',
             '-inf', '']
    cells += [repr(float(x)) for x in rng.standard_normal(300) * 10.0 ** rng.integers(-12, 12, 300)]
    cells += [f'{x:.20f}' for x in rng.random(100)] + [f'{x:.25e}' for x in rng.random(100) * 1e-5]
    n = len(cells)
    for d in ('a', 'b'):
        os.makedirs(tmp_path / d)
        with open(tmp_path / d / 'overview.csv', 'w') as f:
            f.write('read_name,run_id,reverse,saved,l_start_raw,r_end_raw,x\n')
            for i, c in enumerate(cells):
                f.write(f'r{i},0,False,1,{10 + i}.0,{500 + i},{c}\n')
    nat = _hostlib.NativeOverview.open(str(tmp_path / 'a' / 'overview.csv'))
    assert nat is not None, _hostlib.NativeOverview.last_refusal
    path, ref = ov.load_overview(str(tmp_path / 'b'))
    assert nat.lo.tolist() == np.asarray(ref['l_start_raw']).astype(np.int64).tolist()
    ov.store_results(path, ref, [('A', 'AC')] * n, [(0.5, 0.25)] * n, str(tmp_path / 'b'), write=True)
    nat.store(str(tmp_path / 'a'), [1] * n, [2] * n, [0.5] * n, [0.25] * n, np.frombuffer(b'AC' * n, np.uint8), np.arange(n) * 2, write=True)
    assert open(tmp_path / 'a' / 'overview.csv').read() == open(tmp_path / 'b' / 'overview.csv').read()


def test_one_call_setup_equals_the_python_forms(tmp_path):
    """wsh_locus_setup (overview + flank file + both automata + state_similarity.csv in one call) against load_flanks,
    locus_automata and similarity_report, over random loci incl. ones the Python forms raise on (then the library declines)."""
    from warpstr_amd import synth
    from warpstr_amd.caller import similarity_report
    rng = np.random.default_rng(11)
    pm = default_pore_model()
    n_ok = 0
    for trial in range(150):
        pat = random_pattern(rng) if trial % 10 else '(A)' + random_pattern(rng)   # a one-base unit: IndexError in the similarity report
        fl = int(rng.integers(8, 120))
        flanks = [''.join('ACGT'[i] for i in rng.integers(0, 4, size=fl)) for _ in range(4)]
        loc = str(tmp_path / f'l{trial}')
        ov.store_flanks(loc, flanks)
        pd.DataFrame({'read_name': ['a', 'b'], 'run_id': 0, 'reverse': [True, False], 'saved': 1, 'l_start_raw': 0,
                      'r_end_raw': [99, 120]}).to_csv(os.path.join(loc, 'overview.csv'), index=False)
        lim = float(rng.choice([0.75, 0.2, 1.5]))
        st = _hostlib.NativeSetup.run(loc, pat, pm, lim, True)
        assert st is not None and st.overview is not None and st.overview.names == ['a', 'b']
        try:
            want = automata.locus_automata(*ov.load_flanks(loc), pat, pm)
        except Exception:  # noqa: BLE001
            assert st.tables is None
            want = None
        if want is not None:
            assert st.tables is not None
            for got, w in zip(st.tables, want):
                assert 'value' not in got.__dict__   # (nothing copied until somebody looks)
                same_table(got, w)
        try:
            text, warnings, _ = similarity_report(pat, pm, lim)
        except IndexError:
            assert st.similarity is None
            continue
        assert st.similarity == (text, warnings)
        assert open(os.path.join(loc, 'summaries', 'state_similarity.csv')).read() == text
        n_ok += 1
    assert n_ok > 100


def test_hostile_tables_are_declined_or_equal_to_the_pandas_path():
    """scripts/fuzz_overview.py: 800 tables of tokens pandas treats specially; whatever the native parser accepts equals pandas."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'fuzz_overview.py'), '7', '800'], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    last = out.stdout.strip().splitlines()[-1].split()
    assert int(last[1]) > 30 and int(last[3]) > 300 and int(last[5]) == 0, last


def test_complex_unit_tables_equal_the_python_forms(tmp_path):
    """collapse_repeats + store_collapsed in the library against units.collapse_repeats / overview.store_collapsed: random
    patterns (nested units, optional blocks, IUPAC codes, units of the same name), sequences built from their units with errors."""
    from warpstr_amd.loci import _complex_header
    from warpstr_amd.units import break_into_units, collapse_repeats
    rng = np.random.default_rng(21)
    done = 0
    for trial in range(300):
        pat = random_pattern(rng)
        try:
            units, repeat_units, offsets = break_into_units(pat)
        except Exception:  # noqa: BLE001
            continue
        if len(units) < 2:
            continue
        seqs = []
        for _ in range(int(rng.integers(1, 12))):
            s = ''
            for alts, off in zip(repeat_units, offsets):
                s += ''.join('ACGT'[i] for i in rng.integers(0, 4, size=off))
                for _ in range(int(rng.integers(0, 7))):
                    s += alts[int(rng.integers(len(alts)))] if alts and rng.random() < 0.9 else 'ACGT'[int(rng.integers(4))]
            seqs.append(s)
        reverse = [bool(rng.integers(0, 2)) for _ in seqs]
        try:
            want_counts = [collapse_repeats(s, repeat_units, offsets, max_iter=10000) for s in seqs]
        except RuntimeError:  # an empty alternative: the library must leave it to the Python form too
            header = _complex_header(units, repeat_units)
            if header is not None:
                buf = np.frombuffer(''.join(seqs).encode(), np.uint8)
                lens = np.array([len(s) for s in seqs], np.int32)
                assert _hostlib.collapse_store(str(tmp_path), buf, np.cumsum(lens) - lens, lens, reverse, repeat_units, offsets, header[0], header[1], False) is None
            continue
        header = _complex_header(units, repeat_units)
        assert header is not None
        a, b = tmp_path / f'a{trial}', tmp_path / f'b{trial}'
        want = ov.store_collapsed(want_counts, units, repeat_units, reverse, str(b), write=True)
        buf = np.frombuffer(''.join(seqs).encode(), np.uint8)
        lens = np.array([len(s) for s in seqs], np.int32)
        counts, text = _hostlib.collapse_store(str(a), buf, np.cumsum(lens) - lens, lens, reverse, repeat_units, offsets, header[0], header[1], True)
        assert counts.tolist() == [[c for unit in row for c in unit] for row in want_counts]
        rel = os.path.join('predictions', 'complexSTR_analysis', 'complex_repeat_units.csv')
        assert open(a / rel).read() == open(b / rel).read() == text
        import io
        pd.testing.assert_frame_equal(pd.read_csv(io.StringIO(text), index_col=0), want)
        done += 1
    assert done > 60


def test_a_chunk_of_loci_hands_its_columns_over_together(tmp_path):
    """NativeSetup.run_many takes the overview columns of a chunk in two library calls (wsh_loci_counts / wsh_loci_columns); locus
    by locus they must be what NativeOverview.open gives: loci with and without run_id / fast5_path columns, without saved rows,
    with names that are not ASCII, with a table the library declines in the middle of the chunk, in random order."""
    rng = np.random.default_rng(5)
    pm = default_pore_model()
    paths, seqs = [], []
    for i in range(60):
        loc = str(tmp_path / f'l{i}')
        fl = int(rng.integers(8, 40))
        ov.store_flanks(loc, [''.join('ACGT'[k] for k in rng.integers(0, 4, size=fl)) for _ in range(4)])
        n = int(rng.integers(0, 7))
        kind = i % 6
        names = [f'read{i}_{r}' for r in range(n)]
        if kind == 4:
            names = [f'čítanie_{i}_{r}_ž' for r in range(n)]
        df = {'read_name': names, 'reverse': [bool(b) for b in rng.integers(0, 2, size=n)], 'saved': [int(b) for b in (rng.random(n) < 0.8)],
              'l_start_raw': rng.integers(0, 1000, size=n), 'r_end_raw': rng.integers(1000, 9000, size=n)}
        if kind in (0, 2, 4, 5):
            df['run_id'] = [f'run_{r % 2}' for r in range(n)]
        if kind in (0, 3, 4):
            df['fast5_path'] = [f'/data/x/{i}/{r}.fast5' for r in range(n)]
        pd.DataFrame(df).to_csv(os.path.join(loc, 'overview.csv'), index=False)
        if kind == 5 and n:   # a quoted field: the library leaves this table to pandas
            text = open(os.path.join(loc, 'overview.csv')).read().replace('run_0', '"run,0"', 1)
            open(os.path.join(loc, 'overview.csv'), 'w').write(text)
        paths.append(loc)
        seqs.append('(AGC)')
    sts = _hostlib.NativeSetup.run_many(paths, seqs, pm, 0.75, False)
    assert len(sts) == 60
    n_native = 0
    for loc, st in zip(paths, sts):
        one = _hostlib.NativeOverview.open(os.path.join(loc, 'overview.csv'))
        assert (st.overview is None) == (one is None), loc
        if one is None:
            continue
        n_native += 1
        got = st.overview
        assert got.n_saved == one.n_saved and got.n_rows == one.n_rows and got.names == one.names
        assert got.run_id == one.run_id and got.fast5_path == one.fast5_path
        for a, b in ((got.saved, one.saved), (got.reverse, one.reverse), (got.lo, one.lo), (got.hi, one.hi)):
            assert a.dtype == b.dtype and np.array_equal(a, b)
            assert a.flags.writeable and a.base is None   # (its own memory: a locus may change its arrays)
        one.close()
    assert 30 <= n_native < 60   # (the quoted tables and the non-ASCII names go to pandas)


def test_a_chunks_string_columns_are_cut_when_looked_at():
    """_hostlib._Strings: rows [a, b) of a column a chunk of loci handed over as one text -- a sequence of str like the list it
    replaces (index, negative index, slice, iteration, comparison), for a text whose offsets count characters and for a blob with
    non-ASCII names whose offsets count bytes."""
    from warpstr_amd._hostlib import _Strings
    rows = ['read_a', 'b', '', 'čtení-4', 'read_e']
    text = ''.join(rows)
    off = np.cumsum([0] + [len(r) for r in rows]).tolist()
    raw = text.encode('utf-8')
    boff = np.cumsum([0] + [len(r.encode('utf-8')) for r in rows]).tolist()
    for col in (_Strings(text, off, 1, 5), _Strings(raw, boff, 1, 5, True)):
        want = rows[1:5]
        assert len(col) == 4 and list(col) == want and col == want and col[:] == want
        assert [col[k] for k in range(4)] == want and col[-1] == 'read_e' and col[1:3] == want[1:3] and col[::2] == want[::2]
        assert list(map(str, col[0:2])) == want[0:2] and set(col) == set(want)
        with pytest.raises(IndexError):
            col[4]
        with pytest.raises(IndexError):
            col[-5]
    assert len(_Strings(text, off, 2, 2)) == 0 and list(_Strings(text, off, 2, 2)) == []
