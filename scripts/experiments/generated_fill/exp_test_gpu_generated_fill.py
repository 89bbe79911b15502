"""The generated DP fill (warpstr_amd/fillgen.py: a read in four lanes, straight-line code per automaton, compiled at run
time) against the built-in kernels and the oracle: every output of wsx_warp_batch and wsx_call_batch, bit for bit -- ragged
lengths, batch sizes that do not fill a wavefront, reads that are too short, masks, the corner cut, fan-in up to four."""
import numpy as np
import pytest

from oracle import oracle
from tests.helpers import load_case
from warpstr_amd import fillgen, synth
from warpstr_amd.caller import HipCaller, pack_signals

pytestmark = pytest.mark.gpu


def _pair(locus, fl):
    gen = HipCaller([locus.template, locus.reverse], [fl, fl], generated_fill=True)
    ref = HipCaller([locus.template, locus.reverse], [fl, fl], generated_fill=False)
    small = [a for a, t in enumerate((locus.template, locus.reverse)) if t.n_states <= 64]
    assert small and all(isinstance(gen.generated.get(a), dict) and gen.kernel_name(a) == 'wsx_fill_t_u' for a in small), gen.generated
    assert all(ref.kernel_name(a).startswith('dtw_fill_fast<4, ') for a in range(2))
    return gen, ref


@pytest.mark.parametrize('pattern,fl,seed,max_states', [('(AGC)AACAGCCGCCAC(CGC)', 19, 2024, 64), ('(AGC)', 16, 11, None),
                                                        ('(CAGM)', 14, 1, None), ('(CAG)', 28, 5, None), ('(AGC)', 12, 3, None),
                                                        ('((CAGG){CAGM})(CAGA)(CA)', 10, 3, None)])
def test_generated_fill_equals_the_builtin_kernels(pattern, fl, seed, max_states):
    locus = synth.make_locus(pattern, fl, seed, max_states=max_states)
    gen, ref = _pair(locus, fl)  # (the last case: one strand has 67 states and keeps its two-slot kernel, the other is generated)
    rng = np.random.default_rng(seed)
    sigs, revs, _ = synth.batch(locus, 203, (700, 1500), seed + 1, lo=2, hi=12)
    b6 = 6 * (fl - 10)
    for k, at in enumerate(range(60, 90)):  # lengths around the corner cut's thresholds (caller.py:211-224): truncated reads
        sigs[at] = sigs[at][:max(b6 - 6 + k, 12)]
    # degenerate reads in the middle of wavefronts: empty, shorter than min_values_per_state, barely longer
    for at, n in ((5, 0), (17, 3), (40, 4), (41, 5), (100, 9), (150, 1)):
        sigs[at] = rng.normal(size=n)
    sig, off = pack_signals(sigs)
    aut = np.array([int(r) for r in revs], np.int32)
    mask = (rng.random(len(sig)) < 0.5).astype(np.uint8)
    mask = np.repeat(mask[::37], 37)[:len(sig)]  # runs of masked samples, as the bad-repeat mask has them
    for m in (None, mask):
        a, b = gen.warp(sig, off, aut, mask=m, want_last_row=True), ref.warp(sig, off, aut, mask=m, want_last_row=True)
        assert np.array_equal(a['status'], b['status'])
        ok = b['status'] == 0
        assert np.array_equal(a['end_cost'][ok], b['end_cost'][ok])
        for i in np.flatnonzero(ok):
            assert np.array_equal(a['trace'][off[i]:off[i + 1]], b['trace'][off[i]:off[i + 1]]), i
            S = (locus.reverse if aut[i] else locus.template).n_states
            assert np.array_equal(a['last_row'][i, :S], b['last_row'][i, :S]), i
    ra, ea = gen.call(sig, off, aut, want_debug=True, want_seqs=True)
    rb, eb = ref.call(sig, off, aut, want_debug=True, want_seqs=True)
    assert ra.tobytes() == rb.tobytes()
    for i in np.flatnonzero(rb['status'] == 0):
        sl = slice(off[i], off[i + 1])
        for key in ('trace1', 'trace2', 'rescaled', 'badmask'):
            assert np.array_equal(ea[key][sl], eb[key][sl]), (i, key)
        assert np.array_equal(ea['seq2'][off[i]:off[i] + rb['len2'][i]], eb['seq2'][off[i]:off[i] + rb['len2'][i]])
    if fl >= 16:  # (flanks under 16: upstream's IndexError in find_event_borders for every read -- the warp comparison above is what counts there)
        assert int((rb['status'] == 0).sum()) >= 150


@pytest.mark.parametrize('case', ['agc_fl16', 'agc_fl29', 'hd_fl20', 'ngc_fl20', 'agc_fl16_ragged'])
def test_generated_fill_matches_golden(case):
    """The reference-recorded fixtures of the single-slot loci through the generated kernels (test_gpu_parity runs them through
    whatever the handle's default is): paths, lengths and costs as recorded from the upstream caller."""
    from tests.test_gpu_parity import tables_of
    z = load_case(case)
    t, r = tables_of(z)
    if max(t.n_states, r.n_states) > 64:
        pytest.skip('more than 64 states')
    fl = int(z['flank_length'])
    hip = HipCaller([t, r], [fl, fl], generated_fill=True)
    assert hip.kernel_name(0) == 'wsx_fill_t_u' and hip.kernel_name(1) == 'wsx_fill_t_u'
    n = int(z['n_reads'])
    sig, off = pack_signals([z[f'r{i}_signal'] for i in range(n)])
    res, ex = hip.call(sig, off, z['reverse'].astype(np.int32), want_debug=True)
    for i in range(n):
        sl = slice(off[i], off[i + 1])
        assert res['status'][i] == 0
        assert np.array_equal(ex['trace1'][sl], z[f'r{i}_trace1'])
        assert np.array_equal(ex['badmask'][sl], z[f'r{i}_badmask'])
        assert np.array_equal(ex['rescaled'][sl], z[f'r{i}_rescaled'])
        assert np.array_equal(ex['trace2'][sl], z[f'r{i}_trace2'])
        seq, rseq = [str(s) for s in z[f'r{i}_seq']]
        assert (res['len1'][i], res['len2'][i]) == (len(seq), len(rseq))


def test_generated_fill_at_full_size_against_the_oracle():
    """configs[2]'s automaton, 30 000 reads of 2 000 samples (whole wavefronts of equal length: the fast path of the row
    loop), the records of 512 of them against the oracle and all of them against the built-in kernel."""
    import bench
    pattern, fl = bench.HEADLINE
    locus = synth.make_locus(pattern, fl, 2024, max_states=64)
    gen, ref = _pair(locus, fl)
    sigs, revs, _ = synth.batch(locus, 600, 2000, 77, lo=5, hi=25)
    pick = np.random.default_rng(3).integers(0, 600, size=30000)
    rng = np.random.default_rng(4)
    sig = np.concatenate([sigs[i] for i in pick]) + 0.05 * rng.standard_normal(30000 * 2000)
    off = np.arange(30001, dtype=np.int64) * 2000
    aut = np.array([int(revs[i]) for i in pick], np.int32)
    ra, _ = gen.call(sig, off, aut)
    rb, _ = ref.call(sig, off, aut)
    assert ra.tobytes() == rb.tobytes() and int((ra['status'] == 0).sum()) > 29000
    oa = [oracle.Automaton.from_table(locus.template, fl), oracle.Automaton.from_table(locus.reverse, fl)]
    for i in range(512):
        o = oracle.call_read(oa[aut[i]], sig[off[i]:off[i + 1]], debug=False)
        assert (int(ra['status'][i]), int(ra['len1'][i]), int(ra['len2'][i])) == (o.status, o.len1, o.len2)


def test_attach_and_remove_at_run_time():
    """wsx_caller_set_generated_fill(.., NULL) gives the automaton its built-in kernel back; malformed tables are refused."""
    locus = synth.make_locus('(AGC)', 16, 11)
    hip = HipCaller([locus.template, locus.reverse], [16, 16], generated_fill=False)
    sigs, revs, _ = synth.batch(locus, 40, 1200, 5)
    sig, off = pack_signals(sigs)
    aut = np.array([int(r) for r in revs], np.int32)
    base, _ = hip.call(sig, off, aut)
    assert hip.generate_fill(0) and hip.kernel_name(0) == 'wsx_fill_t_u' and hip.kernel_name(1).startswith('dtw_fill_fast')
    mixed, _ = hip.call(sig, off, aut)  # one strand generated, the other built in: two launch groups
    assert mixed.tobytes() == base.tobytes()
    hip.drop_generated_fill(0)
    assert hip.kernel_name(0).startswith('dtw_fill_fast')
    again, _ = hip.call(sig, off, aut)
    assert again.tobytes() == base.tobytes()
    # malformed attachments are refused and leave the handle as it was
    import ctypes as C

    from warpstr_amd import _lib
    gen = fillgen.generate(locus.template, 16)
    code, _ = fillgen.compile_source(gen)
    buf = C.create_string_buffer(code, len(code))
    keep = [np.ascontiguousarray(gen.state_at, np.uint16), np.ascontiguousarray(gen.tb_n, np.uint8),
            np.ascontiguousarray(gen.tb_word, np.uint16), np.ascontiguousarray(gen.tb_pred, np.uint16)]
    bad_pred = keep[3].copy()
    bad_pred[np.flatnonzero(keep[1])[0] * 4] = 999  # a predecessor position outside the automaton
    for fields in ((1, gen.words_per_row + 1, gen.n_per_lane, gen.end_pos, keep),                      # odd row width
                   (1, gen.words_per_row, gen.n_per_lane, (gen.end_pos + 1) % (4 * gen.n_per_lane), keep),   # wrong end position
                   (2, gen.words_per_row, gen.n_per_lane, gen.end_pos, keep),                              # unknown ABI
                   (1, gen.words_per_row, gen.n_per_lane, gen.end_pos, keep[:3] + [bad_pred])):
        g = _lib.WsxGeneratedFill(fields[0], fields[1], fields[2], fields[3], C.cast(buf, C.c_void_p), len(code), *[_lib.ptr(k) for k in fields[4]])
        assert hip.lib.wsx_caller_set_generated_fill(hip.handle, 0, C.byref(g)) == -1 and hip.kernel_name(0).startswith('dtw_fill_fast')
    assert hip.lib.wsx_caller_set_tuning(hip.handle, 99, 1) == -1 and hip.workspace_limit() >= 2 << 30
    hip.set_tuning('generated_passes', 1)  # the unmasked pass generated, the masked one built in: same results
    assert hip.generate_fill(0) and hip.generate_fill(1)
    once, _ = hip.call(sig, off, aut)
    assert once.tobytes() == base.tobytes()
    big = synth.make_locus('(AAAT)', 110, 1)
    assert not fillgen.supported(big.template, 4)
    hip4 = HipCaller([big.template], [110], generated_fill=True)
    assert hip4.generated == {} and hip4.kernel_name(0).startswith('dtw_fill_fast<4, 4')
