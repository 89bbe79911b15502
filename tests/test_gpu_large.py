"""Calls past 2^31 samples: the reference has no such limit (src/caller/caller.py:195-196 -- a read is a NumPy array, a workload a
Python list), the library's host packing, chunk planner, loader and kernels index a call's buffers with 64-bit offsets, and
nothing else exercised that.  Both tests tile a block of 4 096 reads of 30 000 samples eighteen times (2.21 x 10^9 samples): a
32-bit index anywhere makes a later tile differ from the first, or a read near the end of the buffer differ from the oracle."""
import os

import numpy as np
import pytest

from oracle import oracle
from warpstr_amd import _lib, synth
from warpstr_amd.caller import BatchQueue, HipCaller
from warpstr_amd.signal_prep import process_raw

pytestmark = pytest.mark.gpu

BLOCK_READS, READ_SAMPLES, TILES = 4096, 30000, 18
assert BLOCK_READS * READ_SAMPLES * TILES > 2 ** 31


def _locus():
    locus = synth.make_locus('(AGC)', 16, 11)
    oa = [oracle.Automaton.from_table(locus.template, 16), oracle.Automaton.from_table(locus.reverse, 16)]
    return locus, oa


def _room(nbytes):
    import torch
    free, _ = torch.cuda.mem_get_info(0)
    if free < nbytes:
        pytest.skip(f'the device has {free >> 30} GiB free, the test wants {nbytes >> 30} GiB')


def test_a_device_call_past_2_31_samples_with_traces():
    """wsx_call_batch on device buffers whose offsets pass 2^31 (17.7 GB of float64 signal, both state paths requested: 4.4 GB
    each at global offsets): every tile's records and paths equal the first tile's, reads from both ends of the buffer equal
    the oracle (paths, lengths, costs)."""
    import torch
    _room(120 << 30)
    locus, oa = _locus()
    rng = np.random.default_rng(2031)
    dev = torch.device('cuda:0')
    tpl, revs = [], []
    for _ in range(16):
        rev = bool(rng.random() < 0.5)
        tpl.append(synth.squiggle(locus, rev, READ_SAMPLES, rng, lo=300, hi=2000, sigma=0.0)[0])
        revs.append(rev)
    pick = rng.integers(0, len(tpl), size=BLOCK_READS)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    block = torch.from_numpy(np.stack(tpl)).to(dev)[torch.from_numpy(pick).to(dev)]
    block += 0.25 * torch.randn(block.shape, generator=g, device=dev, dtype=torch.float64)
    signal = block.reshape(-1).repeat(TILES)
    n = BLOCK_READS * TILES
    total = n * READ_SAMPLES
    assert signal.numel() == total > 2 ** 31
    offsets = np.arange(n + 1, dtype=np.int64) * READ_SAMPLES
    aut = np.tile(np.array(revs, np.int32)[pick], TILES)
    stream = torch.cuda.current_stream()
    hip = HipCaller([locus.template, locus.reverse], [16, 16], stream=stream.cuda_stream)
    res = torch.zeros((n, _lib.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    tr1 = torch.zeros(total, dtype=torch.int16, device=dev)
    tr2 = torch.zeros(total, dtype=torch.int16, device=dev)
    hip.call_device(signal.data_ptr(), offsets, aut, res.data_ptr(), trace1_ptr=tr1.data_ptr(), trace2_ptr=tr2.data_ptr())
    hip.synchronize()
    torch.cuda.synchronize()
    rec = res.cpu().numpy().view(_lib.RESULT_DTYPE).reshape(TILES, BLOCK_READS)
    assert (rec['status'] == 0).mean() > 0.98
    for t in range(1, TILES):
        assert rec[t].tobytes() == rec[0].tobytes(), f'tile {t}: records differ from the first tile'
    t1, t2 = tr1.view(TILES, -1), tr2.view(TILES, -1)
    for t in range(1, TILES):
        assert torch.equal(t1[t], t1[0]) and torch.equal(t2[t], t2[0]), f'tile {t}: state paths differ from the first tile'
    flat = rec.reshape(-1)
    for i in (0, 1, BLOCK_READS - 1, n - BLOCK_READS, n - 2, n - 1):
        a, b = int(offsets[i]), int(offsets[i + 1])
        o = oracle.call_read(oa[aut[i]], signal[a:b].cpu().numpy())
        assert int(flat['status'][i]) == o.status
        if o.status == 0:
            assert np.array_equal(tr1[a:b].cpu().numpy().view(np.uint16), o.trace1), i
            assert np.array_equal(tr2[a:b].cpu().numpy().view(np.uint16), o.trace2), i
            assert (int(flat['len1'][i]), int(flat['len2'][i])) == (o.len1, o.len2)
            assert abs(flat['cost2'][i] - o.cost2) <= 1e-5 * abs(o.cost2)
    hip.close()


def test_raw_reads_past_2_31_samples_through_submit_raw_parts(tmp_path):
    """The loader in front of the caller with raw offsets past 2^31: a batch of 73 728 raw int16 reads of 30 000 samples
    (4.4 GB in HBM), handed over as eighteen parts of one reader arena (BatchQueue.submit_raw_parts), an STR segment of 2 500
    samples inside every read; every part's records equal the first part's, reads from both ends equal the host restatement of
    the loader + the oracle."""
    import torch
    _room(60 << 30)
    shm = '/dev/shm' if os.path.isdir('/dev/shm') and os.access('/dev/shm', os.W_OK) else str(tmp_path)
    st = os.statvfs(shm)
    if st.f_bavail * st.f_frsize < (BLOCK_READS * READ_SAMPLES * 2) + (64 << 20):
        pytest.skip('no room for a 246 MB arena')
    locus, oa = _locus()
    rng = np.random.default_rng(2032)
    seg = 2500
    tpl, revs = [], []
    for _ in range(16):
        rev = bool(rng.random() < 0.5)
        tpl.append(synth.squiggle(locus, rev, seg, rng, lo=5, hi=40, sigma=0.0)[0])
        revs.append(rev)
    pick = rng.integers(0, len(tpl), size=BLOCK_READS)
    lo_b = 9000 + 37 * (np.arange(BLOCK_READS) % 113)
    hi_b = lo_b + seg - 1
    x = rng.normal(0.0, 1.0, size=(BLOCK_READS, READ_SAMPLES))
    for i in range(BLOCK_READS):
        x[i, lo_b[i]:hi_b[i] + 1] = tpl[pick[i]] + 0.25 * rng.standard_normal(seg)
    raw = np.clip(np.round(x * 70.0 + 500.0), 0, 2047).astype(np.int16)
    del x
    path = os.path.join(shm, f'warpstr_test_arena_{os.getpid()}')
    try:
        raw.tofile(path)
        cap = raw.size
        n = BLOCK_READS * TILES
        parts = [(path, cap, 0, [READ_SAMPLES] * BLOCK_READS) for _ in range(TILES)]
        lo, hi = np.tile(lo_b, TILES).astype(np.int64), np.tile(hi_b, TILES).astype(np.int64)
        aut = np.tile(np.array(revs, np.int32)[pick], TILES)
        stream = torch.cuda.Stream(device=torch.device('cuda:0'))
        hip = HipCaller([locus.template, locus.reverse], [16, 16], stream=stream.cuda_stream)
        queue = BatchQueue(hip, stream)
        ticket = queue.submit_raw_parts(0, parts, lo, hi, aut)
        rec, s1, p1, s2, p2 = queue.collect(ticket)
        assert len(rec) == n and int(np.sum([READ_SAMPLES] * n)) > 2 ** 31
        tiles = rec.reshape(TILES, BLOCK_READS)
        assert (rec['status'] == 0).mean() > 0.98
        for t in range(1, TILES):
            assert tiles[t].tobytes() == tiles[0].tobytes(), f'part {t}: records differ from the first part'
        # the called sequences come down packed: every part's bytes equal the first part's
        per = int(p2[BLOCK_READS])
        assert int(p2[-1]) == per * TILES
        s2t = s2.reshape(TILES, per)
        assert all(np.array_equal(s2t[t], s2t[0]) for t in range(1, TILES))
        for i in (0, 1, BLOCK_READS - 1, n - BLOCK_READS, n - 2, n - 1):
            r = i % BLOCK_READS
            o = oracle.call_read(oa[aut[i]], process_raw(raw[r], (int(lo[i]), int(hi[i])), 'Brute'), debug=False)
            assert int(rec['status'][i]) == o.status
            if o.status == 0:
                assert (int(rec['len1'][i]), int(rec['len2'][i])) == (o.len1, o.len2)
                assert abs(rec['cost1'][i] - o.cost1) <= 1e-5 * abs(o.cost1) and abs(rec['cost2'][i] - o.cost2) <= 1e-5 * abs(o.cost2)
        hip.synchronize()
        queue.close()
        hip.close()
    finally:
        if os.path.exists(path):
            os.unlink(path)
