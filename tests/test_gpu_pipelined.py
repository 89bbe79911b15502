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
