"""Pipelined device-buffer calls (wsx_caller_set_pipelined / wsx_caller_join): same bytes as stream-ordered calls."""
import numpy as np
import pytest

from warpstr_amd import _lib, synth
from warpstr_amd.caller import HipCaller, pack_signals

pytestmark = pytest.mark.gpu


def _batches(locus):
    out = []
    # growing sizes: the work sets are re-allocated while earlier calls are still in flight
    for seed, n, T in ((3, 300, (400, 2500)), (2, 4200, 800), (1, 8500, (500, 1100)), (4, 8200, 600)):
        sigs, revs, _ = synth.batch(locus, n, T, seed)
        sig, off = pack_signals(sigs)
        out.append((sig, off, np.array([1 if x else 0 for x in revs], dtype=np.int32)))
    return out


def test_pipelined_calls_match_stream_ordered_calls():
    import torch
    locus = synth.make_locus('(AGC)AACAGCCGCCAC(CGC)', 19, 2024, max_states=64)
    batches = _batches(locus)
    hip = HipCaller([locus.template, locus.reverse], [19, 19], stream=torch.cuda.current_stream().cuda_stream)
    ref = HipCaller([locus.template, locus.reverse], [19, 19])  # its own handle: `hip` starts the pipelined calls cold
    want = [ref.call(sig, off, aut, want_traces=True) for sig, off, aut in batches]
    dev = torch.device('cuda:0')
    d_sig = [torch.from_numpy(sig).to(dev) for sig, _, _ in batches]

    def outputs():
        return [(torch.zeros((len(aut), _lib.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev),
                 torch.zeros(len(sig), dtype=torch.int16, device=dev)) for sig, _, aut in batches]

    def check(outs):
        for (res, tr2), (w_res, w_extra) in zip(outs, want):
            got = res.cpu().numpy().view(_lib.RESULT_DTYPE).reshape(-1)
            assert got.tobytes() == w_res.tobytes()
            assert np.array_equal(tr2.cpu().numpy().view(np.uint16), w_extra['trace2'])

    hip.set_pipelined(True)
    rounds = [outputs() for _ in range(3)]  # 12 calls back to back, four of them in every slot parity
    for outs in rounds:
        for (sig, off, aut), ds, (res, tr2) in zip(batches, d_sig, outs):
            hip.call_device(ds.data_ptr(), off, aut, res.data_ptr(), trace2_ptr=tr2.data_ptr())
    # (a) a side stream ordered after the calls with join() sees finished outputs
    side = torch.cuda.Stream(device=dev)
    hip.join(side.cuda_stream)
    with torch.cuda.stream(side):
        copies = [[(res.clone(), tr2.clone()) for res, tr2 in outs] for outs in rounds]
    side.synchronize()
    for outs in copies:
        check(outs)
    # (b) host-side wait
    hip.synchronize()
    for outs in rounds:
        check(outs)
    # (c) a host-buffer call in between stays synchronous and correct
    r, e = hip.call(*batches[1], want_traces=True)
    assert r.tobytes() == want[1][0].tobytes() and np.array_equal(e['trace2'], want[1][1]['trace2'])
    # (d) turning the mode off restores stream order: the handle's (= torch's current) stream is enough
    outs = outputs()
    (sig, off, aut), ds, (res, tr2) = batches[0], d_sig[0], outs[0]
    hip.call_device(ds.data_ptr(), off, aut, res.data_ptr(), trace2_ptr=tr2.data_ptr())
    hip.set_pipelined(False)
    (sig, off, aut), ds, (res, tr2) = batches[3], d_sig[3], outs[3]
    hip.call_device(ds.data_ptr(), off, aut, res.data_ptr(), trace2_ptr=tr2.data_ptr())
    torch.cuda.current_stream().synchronize()
    for k in (0, 3):
        got = outs[k][0].cpu().numpy().view(_lib.RESULT_DTYPE).reshape(-1)
        assert got.tobytes() == want[k][0].tobytes()
        assert np.array_equal(outs[k][1].cpu().numpy().view(np.uint16), want[k][1]['trace2'])


def test_join_before_any_call_is_a_no_op():
    import torch
    locus = synth.make_locus('(AGC)', 16, 1)
    hip = HipCaller([locus.template, locus.reverse], [16, 16], stream=torch.cuda.current_stream().cuda_stream)
    hip.join()
    hip.set_pipelined(True)
    hip.join()
    hip.synchronize()
    hip.set_pipelined(False)


def test_timing_window_covers_every_call():
    import torch
    locus = synth.make_locus('(AGC)', 16, 1)
    sigs, revs, _ = synth.batch(locus, 9000, 700, 3)
    sig, off = pack_signals(sigs)
    aut = np.array([1 if x else 0 for x in revs], dtype=np.int32)
    dev = torch.device('cuda:0')
    dsig = torch.from_numpy(sig).to(dev)
    res = torch.zeros((len(aut), _lib.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    hip = HipCaller([locus.template, locus.reverse], [16, 16], stream=torch.cuda.current_stream().cuda_stream)
    hip.set_pipelined(True)  # (a small pipelined call is cut into fewer chunks than a stream-ordered one)
    hip.call_device(dsig.data_ptr(), off, aut, res.data_ptr())
    one = hip.last_timing()
    assert one['dp_launches'] >= 2 and one['dp_kernel_ms'] > 0 and one['total_ms'] >= one['dp_kernel_ms'] / one['dp_launches']
    b, e, r = hip.fill_intervals()
    assert len(b) == one['dp_launches'] and (e > b).all() and r.sum() == 2 * len(aut)  # two passes over every read
    hip.timing_window(True)
    for _ in range(5):
        hip.call_device(dsig.data_ptr(), off, aut, res.data_ptr())
    five = hip.last_timing()
    assert five['dp_launches'] == 5 * one['dp_launches']
    assert five['dp_kernel_ms'] > 2.5 * one['dp_kernel_ms'] and five['total_ms'] > 2.5 * one['total_ms']
    hip.timing_window(False)
    hip.call_device(dsig.data_ptr(), off, aut, res.data_ptr())
    again = hip.last_timing()
    assert again['dp_launches'] == one['dp_launches']


def test_pipelined_calls_with_fitpacks_smoothing_branch():
    """rescaling.threshold > 1: the one stage with a host decision in the middle of a chunk (how many reads take FITPACK's
    knot-adding branch sizes their workspace).  Pipelined device-buffer calls on several streams, chunks with and without
    such reads, growing counts (the workspace is re-allocated while other calls are in flight): same bytes as the
    synchronous host-buffer calls of another handle."""
    import copy

    import torch
    from warpstr_amd.caller import RescalerConfig
    base = synth.make_locus('(AGC)AACAGCCGCCAC(CGC)', 20, 31)
    locus = copy.deepcopy(base)
    for t in (locus.template, locus.reverse):
        t.value = np.where(np.arange(t.n_states) % 2 == 0, 3.5, -3.5).astype(np.float64)
    rc = RescalerConfig(threshold=6.0, max_std=3.0)
    rng = np.random.default_rng(12)
    batches = []
    for n, noise in ((40, 0.05), (300, 0.3), (1200, 0.05), (700, 1.0)):
        sigs = [rng.normal(0.0, noise, size=int(rng.integers(900, 2200))) for _ in range(n)]
        sig, off = pack_signals(sigs)
        batches.append((sig, off, (rng.random(n) < 0.5).astype(np.int32)))
    hip = HipCaller([locus.template, locus.reverse], [20, 20], rescaler_config=rc, stream=torch.cuda.current_stream().cuda_stream)
    ref = HipCaller([locus.template, locus.reverse], [20, 20], rescaler_config=rc)
    want = [ref.call(sig, off, aut, want_debug=True) for sig, off, aut in batches]
    assert sum(int((w[0]['status'] == 0).sum()) for w in want) > 1000
    # ... and the batches do hold reads of that kind (the oracle says which: more than the cubic's eight knots)
    from oracle import oracle
    oa = [oracle.Automaton.from_table(locus.template, 20), oracle.Automaton.from_table(locus.reverse, 20)]
    prm = oracle.Params(threshold=6.0, max_std=3.0)
    sig0, off0, aut0 = batches[0]
    knots = [oracle.call_read(oa[aut0[i]], sig0[off0[i]:off0[i + 1]], prm) for i in range(12)]
    assert sum(1 for o in knots if o.status == 0 and o.fit_knots > 8) >= 4
    for i, o in enumerate(knots):
        assert want[0][0]['status'][i] == o.status
        if o.status == 0:
            assert np.array_equal(want[0][1]['trace2'][off0[i]:off0[i + 1]], o.trace2)
    dev = torch.device('cuda:0')
    d_sig = [torch.from_numpy(sig).to(dev) for sig, _, _ in batches]
    hip.set_pipelined(True)
    rounds = []
    for _ in range(3):
        outs = [(torch.zeros((len(aut), _lib.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev),
                 torch.zeros(len(sig), dtype=torch.int16, device=dev)) for sig, _, aut in batches]
        for (sig, off, aut), ds, (res, tr2) in zip(batches, d_sig, outs):
            hip.call_device(ds.data_ptr(), off, aut, res.data_ptr(), trace2_ptr=tr2.data_ptr())
        rounds.append(outs)
    hip.synchronize()
    for outs in rounds:
        for (res, tr2), (w_res, w_extra) in zip(outs, want):
            got = res.cpu().numpy().view(_lib.RESULT_DTYPE).reshape(-1)
            assert got.tobytes() == w_res.tobytes()
            assert np.array_equal(tr2.cpu().numpy().view(np.uint16), w_extra['trace2'])


def test_automata_added_while_calls_are_in_flight():
    """wsx_caller_add_automata: a handle created with one locus takes three more -- a bigger one (more slots: another kernel
    variant, more states than any automaton before), a single-slot one, and a copy of the first -- while pipelined calls that name
    the earlier automata are still running on the handle's streams.  Every call's records and second state paths equal those of
    a handle that was created with all eight automata at once, and a malformed addition leaves the handle as it was."""
    import torch
    loci = [synth.make_locus('(AGC)AACAGCCGCCAC(CGC)', 19, 2024, max_states=64), synth.make_locus('(AAAT)', 110, 5),
            synth.make_locus('(AGC)', 16, 11), synth.make_locus('(AGC)AACAGCCGCCAC(CGC)', 19, 2024, max_states=64)]
    flanks = [19, 110, 16, 19]
    dev = torch.device('cuda:0')
    stream = torch.cuda.current_stream()

    def batch(li, n, T, seed):
        sigs, revs, _ = synth.batch(loci[li], n, T, seed, lo=3, hi=12)
        sig, off = pack_signals(sigs)
        return torch.from_numpy(sig).to(dev), off, np.array([2 * li + (1 if x else 0) for x in revs], dtype=np.int32)
    # calls in the order they are enqueued; `after` = how many loci the handle must hold before the call
    plan = [(0, 12000, 900, 1, 1), (0, 9000, 1100, 2, 1), (1, 3000, (2300, 3200), 3, 2), (0, 6000, 800, 4, 2), (2, 5000, 1200, 5, 3),
            (1, 2000, (2300, 3000), 6, 3), (3, 7000, 900, 7, 4), (0, 4000, 700, 8, 4)]
    work = [batch(li, n, T, seed) + (after,) for li, n, T, seed, after in plan]
    all_tables = [t for l in loci for t in (l.template, l.reverse)]
    ref = HipCaller(all_tables, [f for f in flanks for _ in range(2)], stream=stream.cuda_stream)
    want = []
    for sig, off, aut, _ in work:
        res = torch.zeros((len(aut), _lib.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        tr2 = torch.zeros(len(sig), dtype=torch.int16, device=dev)
        ref.call_device(sig.data_ptr(), off, aut, res.data_ptr(), trace2_ptr=tr2.data_ptr())
        ref.synchronize()
        want.append((res.cpu().numpy().tobytes(), tr2.cpu().numpy()))
    names_ref = [ref.kernel_name(a) for a in range(8)]
    ref.close()

    hip = HipCaller([loci[0].template, loci[0].reverse], [19, 19], stream=stream.cuda_stream)
    hip.set_pipelined(True)
    held = 1
    outs = []
    for sig, off, aut, after in work:
        while held < after:   # (the earlier calls are still on the device: nothing is waited for)
            first = hip.add_automata([loci[held].template, loci[held].reverse], [flanks[held]] * 2)
            assert first == 2 * held
            held += 1
        res = torch.zeros((len(aut), _lib.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        tr2 = torch.zeros(len(sig), dtype=torch.int16, device=dev)
        hip.call_device(sig.data_ptr(), off, aut, res.data_ptr(), trace2_ptr=tr2.data_ptr())
        outs.append((res, tr2))
    # a malformed addition: refused, the handle unchanged
    bad = synth.make_locus('(AGC)', 16, 11).template
    import copy
    bad = copy.copy(bad)
    bad.pred_idx = np.where(np.arange(len(bad.pred_idx)) == 3, 10 ** 6, bad.pred_idx).astype(np.int32)
    with pytest.raises(RuntimeError, match='wsx_caller_add_automata'):
        hip.add_automata([bad], [16], [False])
    assert len(hip.automata) == 8
    hip.synchronize()
    torch.cuda.synchronize()
    for k, ((res, tr2), (w_res, w_tr2)) in enumerate(zip(outs, want)):
        assert res.cpu().numpy().tobytes() == w_res, f'call {k}: records differ from the handle created with every automaton'
        assert np.array_equal(tr2.cpu().numpy(), w_tr2), f'call {k}: state paths differ'
    assert [hip.kernel_name(a) for a in range(8)] == names_ref
    assert len({hip.kernel_name(a) for a in range(8)}) >= 3   # single-slot packed, lane-major four-slot, single-slot plain
    # ... and a call after the refused addition still works
    sig, off, aut, _ = work[0]
    res = torch.zeros((len(aut), _lib.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    hip.call_device(sig.data_ptr(), off, aut, res.data_ptr())
    hip.synchronize()
    assert res.cpu().numpy().tobytes() == want[0][0]
    hip.close()
