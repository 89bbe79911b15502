"""The generator of the per-automaton DP fill (warpstr_amd/fillgen.py) on the CPU: its tables describe the automaton it was
given, its source is valid HIP for gfx950 (hipcc cross-compiles without a GPU), the cache works.  What the generated kernels
COMPUTE is checked on the GPU (tests/exp_test_gpu_generated_fill.py)."""
import os
import shutil

import numpy as np
import pytest

from warpstr_amd import fillgen, synth

CASES = [('(AGC)AACAGCCGCCAC(CGC)', 19, 2024, 64), ('(AGC)', 16, 11, None), ('(CAGM)', 14, 1, None), ('(CAG)', 28, 5, None)]


@pytest.mark.parametrize('pattern,fl,seed,max_states', CASES)
def test_tables_describe_the_automaton(pattern, fl, seed, max_states):
    locus = synth.make_locus(pattern, fl, seed, max_states=max_states)
    for t in (locus.template, locus.reverse):
        assert fillgen.supported(t, 4) and not fillgen.supported(t, 5)
        g = fillgen.generate(t, fl)
        S, n = t.n_states, g.n_per_lane
        P = 4 * n
        assert P >= S and g.words_per_row % 2 == 0 and 2 <= g.words_per_row <= 32
        pos_of = {int(j): p for p, j in enumerate(g.state_at) if j != 0xFFFF}
        assert sorted(pos_of) == list(range(S)) and len(g.state_at) == P            # every state has exactly one position
        assert g.state_at[g.end_pos] == t.endstate
        tb_word, tb_pred = g.tb_word.reshape(P, fillgen.MAX_F), g.tb_pred.reshape(P, fillgen.MAX_F)
        used = set()
        for j in range(S):
            p, inc = pos_of[j], t.incoming(j)
            assert g.tb_n[p] == len(inc)
            for f, pred in enumerate(inc):                                          # candidates in `incoming` order
                assert g.state_at[tb_pred[p, f]] == pred and tb_word[p, f] < g.words_per_row
                used.add((int(tb_word[p, f]), p // n))
        # a word's bit for lane q belongs to one (state, candidate) only
        assert len(used) == sum(len(t.incoming(j)) for j in range(S))
        assert all(g.tb_n[p] == 0 for p in range(P) if g.state_at[p] == 0xFFFF)
        # chain states sit next to their predecessor: most groups are the chain's (that is the point of the layout)
        chain = sum(1 for j in range(S) if len(t.incoming(j)) == 1 and pos_of[t.incoming(j)[0]] == pos_of[j] - 1)
        assert chain >= S // 2  # (IUPAC alternatives and loop entries are the rest)
        assert g.valu_per_wave_row <= 11 * 16                                        # <= 11 per read-row by construction
        assert 'wsx_fill_t_u' in g.source and 'wsx_fill_t_m' in g.source and len(g.key) == 24


def test_unsupported_automata_are_refused():
    big = synth.make_locus('(AAAT)', 110, 1)
    assert not fillgen.supported(big.template, 4)
    ngc = synth.make_locus('(NGC)', 16, 1)  # > 64 states
    assert not fillgen.supported(ngc.template, 4)


def test_source_compiles_for_gfx950_and_is_cached(tmp_path, monkeypatch):
    """`hipcc --genco` here (no GPU needed); the second request is served from the cache without a compiler."""
    if not os.path.exists(os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')):
        pytest.skip('no hipcc')
    monkeypatch.setenv('WARPSTR_CACHE_DIR', str(tmp_path))
    monkeypatch.setenv('WARPSTR_FILLGEN_HIPCC', '1')
    locus = synth.make_locus('(AGC)', 16, 11)
    g = fillgen.generate(locus.template, 16)
    assert fillgen.compile_source(g, compile_missing=False) == (None, 'not in the cache')
    code, how = fillgen.compile_source(g)
    assert how == 'hipcc' and (code[:4] == b'\x7fELF' or code.startswith(b'__CLANG_OFFLOAD_BUNDLE__')) and len(code) > 20000
    assert os.path.exists(os.path.join(str(tmp_path), g.key + '.hsaco'))
    monkeypatch.setenv('HIPCC', '/nonexistent/hipcc')
    again, how = fillgen.compile_source(g)
    assert how == 'cache' and again == code
    # another automaton is another source and another key; generator options are part of the key
    g2 = fillgen.generate(locus.reverse, 16)
    monkeypatch.setenv('WARPSTR_FILLGEN_OPTS', 'sb=0')
    g3 = fillgen.generate(locus.template, 16)
    assert len({g.key, g2.key, g3.key}) == 3
    shutil.rmtree(str(tmp_path), ignore_errors=True)
