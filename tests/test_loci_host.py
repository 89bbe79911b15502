"""Host logic of the several-loci step-3 driver (warpstr_amd/loci.py) on CPU: batching, cost sharding, the two all-gathers,
per-locus outputs, error agreement across ranks -- with a stand-in for the GPU engine (a deterministic function of each
read's samples; the product has no CPU engine).  world_size 1, 2 and 8 over gloo."""
import filecmp
import os
import socket

import numpy as np
import pandas as pd
import pytest
import torch.multiprocessing as mp

from warpstr_amd import _lib, overview as ov, synth
from warpstr_amd.caller import ReadCallError, ragged_index, slice_lengths
from warpstr_amd.wrapper import LocusPath, main_wrapper_loci

OUTPUTS = ['overview.csv', 'predictions/sequences/all.fasta', 'predictions/sequences/sequences_template.fasta',
           'predictions/sequences/sequences_reverse.fasta', 'summaries/state_similarity.csv']
LOCI = [('(AGC)', 16, 7), ('(AGC)AACAGCCGCCAC(CGC)', 20, 12), ('(AAAT)', 30, 1), ('(GGCCCC)', 24, 0), ('(CAG)CAACAG(CCG)', 20, 9)]


def _free_port():
    from tests.helpers import free_port
    return free_port()


class FakeEngine:
    """Same four methods as loci.HipEngine; a read's record and sequences are a function of its samples and automaton."""

    def __init__(self, tables, flank_lengths, caller_config, rescaler_config, device):
        self.n_aut = len(tables)
        self.key = [int(t.n_states) + int(t.endstate) for t in tables]   # (a property of the automaton, not its place in the handle)
        self.batches = 0

    def add_automata(self, tables, flank_lengths):
        first = self.n_aut
        self.key += [int(t.n_states) + int(t.endstate) for t in tables]
        self.n_aut += len(tables)
        self.added = getattr(self, 'added', 0) + len(tables)
        return first

    def submit_signals(self, signals, aut):
        self.batches += 1
        n = len(signals)
        rec = np.zeros(n, dtype=_lib.RESULT_DTYPE)
        s1, s2 = [], []
        for i, (x, a) in enumerate(zip(signals, aut)):
            assert 0 <= a < self.n_aut
            h = int(abs(float(np.sum(x))) * 1000) + self.key[a]
            rec['status'][i] = 3 if x[0] > 90.0 else 0
            rec['len1'][i], rec['len2'][i] = 4 + h % 9, 3 + h % 11
            rec['cost1'][i], rec['cost2'][i] = float(np.mean(x)), float(np.max(x))
            if rec['status'][i] == 0:
                s1.append(('ACGT' * 5)[h % 4:][:rec['len1'][i]])
                s2.append(('TTGCA' * 5)[h % 5:][:rec['len2'][i]])
            else:
                s1.append('')
                s2.append('')
        return rec, s1, s2

    def submit_raw(self, raws, lo, hi, aut):
        return self.submit_signals([np.asarray(r[a:b + 1], np.float64) / 64.0 for r, a, b in zip(raws, lo, hi)], aut)

    def collect(self, ticket):
        rec, s1, s2 = ticket
        pos = lambda ss: np.concatenate([[0], np.cumsum([len(x) for x in ss])]).astype(np.int64)
        u8 = lambda ss: np.frombuffer(''.join(ss).encode(), np.uint8)
        return rec, u8(s1), pos(s1), u8(s2), pos(s2)

    def info(self):
        return {'batches': self.batches}

    def close(self):
        pass


class SharedFakeEngine(FakeEngine):
    """... with the staging buffers the reader processes decode into (caller.SharedStaging; nothing to page-lock on a CPU)."""

    def __init__(self, *a):
        super().__init__(*a)
        from warpstr_amd.caller import SharedStaging
        self.staging = SharedStaging(3)
        self.shared_batches = 0

    def stage_shared(self, count):
        return self.staging.take(count)

    def stage_local(self, count):
        self.local_batches = getattr(self, 'local_batches', 0) + 1
        return dict(path=None, view=np.zeros(max(count, 1), np.int16), event=None)

    def submit_raw_shared(self, slot, roff, lo, hi, aut):
        self.shared_batches += 1
        raws = [slot['view'][roff[r]:roff[r + 1]].copy() for r in range(len(roff) - 1)]
        return self.submit_raw(raws, lo, hi, aut)

    def info(self):
        return {'batches': self.batches, 'shared_batches': self.shared_batches, 'local_batches': getattr(self, 'local_batches', 0)}


class ArenaFakeEngine(SharedFakeEngine):
    """... and with the reader arenas (every reader process decodes into files of its own, one upload per chunk)."""
    ARENA_REGIONS = 3

    def region_wait(self, region):
        pass

    def submit_raw_parts(self, region, parts, lo, hi, aut):
        self.arena_batches = getattr(self, 'arena_batches', 0) + 1
        raws = []
        for path, cap, base, lens in parts:
            view = np.memmap(path, dtype=np.int16, mode='r')
            at = base
            for n in lens:
                raws.append(np.array(view[at:at + n]))
                at += n
        return self.submit_raw(raws, lo, hi, aut)

    def info(self):
        return dict(super().info(), arena_batches=getattr(self, 'arena_batches', 0))

    def close(self):
        self.staging.close()


class VbzFakeEngine(ArenaFakeEngine):
    """... and one that takes the reads as the blocks inside their VBZ chunks (wsx_vbz_decode's part played by oracle/vbz.py)."""
    DEVICE_ZSTD = False

    def submit_vbz_parts(self, region, parts, lo, hi, aut):
        from oracle import vbz
        self.vbz_batches = getattr(self, 'vbz_batches', 0) + 1
        raws = []
        for path, cap, base, used, lens, table in parts:
            view = np.memmap(path, dtype=np.uint8, mode='r')
            assert cap <= len(view) and base + used <= cap
            t = np.frombuffer(table, np.int64).reshape(-1, 7)
            per_read = [[] for _ in lens]
            for r, kind, off, nbytes, ns, nv, content in t:
                assert off % 16 == 0 and base <= off and off + nbytes <= base + used and nv >= ns
                blk = np.array(view[off:off + nbytes])
                if kind >= 3:   # the chunk's zstd frame, left for the device: wsx_zstd_decode's part played by oracle/zstd_oracle.c
                    from oracle import zstd as ozstd
                    self.zstd_frames = getattr(self, 'zstd_frames', 0) + 1
                    assert self.DEVICE_ZSTD and ozstd.content_size(blk.tobytes()) == content
                    blk, kind = np.frombuffer(ozstd.decode(blk.tobytes()), np.uint8), kind - 2
                else:
                    assert content == 0
                per_read[r].append(blk.view(np.int16)[:ns] if kind == 0 else vbz.decode_block(blk, int(nv), kind == 1)[:ns])
            for n, pieces in zip(lens, per_read):
                raw = np.concatenate(pieces)
                assert len(raw) == n
                raws.append(raw)
        return self.submit_raw(raws, lo, hi, aut)

    def info(self):
        return dict(super().info(), vbz_batches=getattr(self, 'vbz_batches', 0), zstd_frames=getattr(self, 'zstd_frames', 0))


def _make_loci(root, poison=None):
    """Five synthetic locus directories (one without saved reads, one with a single read) + the normalised segments by read
    name; returns (loci, {read name: signal})."""
    signals, loci = {}, []
    for li, (pattern, fl, n) in enumerate(LOCI):
        locus = synth.make_locus(pattern, fl, 40 + li)
        sigs, revs, _ = synth.batch(locus, n, (600, 1400), 50 + li, lo=3, hi=9)
        loc = os.path.join(root, f'locus{li}')
        ov.store_flanks(loc, [locus.left_t, locus.right_t, locus.left_r, locus.right_r])
        names = [f'L{li}_read{i:03d}' for i in range(n + 2)]
        lens = [len(s) for s in sigs] + [100, 100]
        pd.DataFrame({'read_name': names, 'run_id': 0, 'reverse': list(revs) + [False, True], 'saved': [1] * n + [0, 0],
                      'l_start_raw': 1000, 'r_end_raw': [1000 + L - 1 for L in lens]}).to_csv(os.path.join(loc, 'overview.csv'), index=False)
        signals.update({nm: s for nm, s in zip(names, sigs)})
        loci.append(LocusPath(loc, pattern, fl))
    if poison:
        signals[poison] = signals[poison].copy()
        signals[poison][0] = 99.0
    return loci, signals


def _loader(signals, fail_on=None):
    def load(fast5path, lo, hi):
        name = os.path.basename(fast5path)[:-len('.fast5')]
        if name == fail_on:
            raise FileNotFoundError(f'{fast5path} is missing')
        return signals[name]
    return load


def test_ragged_index_and_slice_lengths():
    starts, lens = np.array([5, 0, 20, 9]), np.array([3, 0, 2, 1])
    assert ragged_index(starts, lens).tolist() == [5, 6, 7, 20, 21, 9]
    assert ragged_index(starts, np.zeros(4, int)).tolist() == []
    rng = np.random.default_rng(0)
    L = rng.integers(0, 50, 200)
    lo, hi = rng.integers(0, 60, 200), rng.integers(-1, 70, 200)
    want = [len(np.arange(n)[a:b + 1]) for a, b, n in zip(lo, hi, L)]
    assert slice_lengths(lo, hi, L).tolist() == want
    lo[::7] -= 30  # negative starts count from the end, as Python slices do
    want = [len(np.arange(n)[a:b + 1]) for a, b, n in zip(lo, hi, L)]
    assert slice_lengths(lo, hi, L).tolist() == want


def test_one_rank_small_batches_equal_one_batch(tmp_path):
    """The cut into batches does not show in the outputs (batch_reads = 4 against everything in one batch), and a locus
    without saved reads / with one read gets its files like any other."""
    a, sig = _make_loci(str(tmp_path / 'a'))
    b, _ = _make_loci(str(tmp_path / 'b'))
    tm = {}
    ta = main_wrapper_loci(a, 1, signal_loader=_loader(sig), _engine=FakeEngine, quiet=True, batch_reads=4, timings=tm)
    tb = main_wrapper_loci(b, 1, signal_loader=_loader(sig), _engine=FakeEngine, quiet=True)
    assert tm['batches'] >= 7 and tm['n_loci'] == 5 and tm['n_reads'] == 29
    for la, lb, (dfa, _), (dfb, _) in zip(a, b, ta, tb):
        for rel in OUTPUTS:
            assert filecmp.cmp(os.path.join(la.path, rel), os.path.join(lb.path, rel), shallow=False), rel
        assert dfa.equals(dfb)
    assert os.path.exists(os.path.join(a[1].path, 'predictions/complexSTR_analysis/complex_repeat_units.csv'))
    df = pd.read_csv(os.path.join(a[3].path, 'overview.csv'))
    assert (df['results'] == -1).all()  # nothing saved: every row -1, as upstream writes it


def _rank(rank, world, port, root, mode, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    loci, sig = _make_loci(root) if rank == 0 else (None, None)
    dist.barrier()
    if rank != 0:  # the directories exist now; every rank derives the same signals
        import tempfile
        with tempfile.TemporaryDirectory() as scratch:
            _, sig = _make_loci(scratch)
        loci = [LocusPath(os.path.join(root, f'locus{li}'), p, fl) for li, (p, fl, _) in enumerate(LOCI)]
    if mode == 'poison':
        sig['L1_read004'] = sig['L1_read004'].copy()
        sig['L1_read004'][0] = 99.0
    msg = 'ok'
    try:
        main_wrapper_loci(loci, 1, signal_loader=_loader(sig, 'L4_read002' if mode == 'missing' else None), _engine=FakeEngine,
                          quiet=True, shard=True, batch_reads=5)
    except Exception as e:  # noqa: BLE001
        msg = f'{type(e).__name__}: {e}'
    with open(os.path.join(out_dir, f'rank{rank}.txt'), 'w') as f:
        f.write(msg)
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 8])
def test_ranks_write_what_one_rank_writes(tmp_path, world):
    """world 2 and world 8 (29 reads: shards of 3-4 reads, the single-read locus on one rank) against one rank."""
    one, sig = _make_loci(str(tmp_path / 'one'))
    main_wrapper_loci(one, 1, signal_loader=_loader(sig), _engine=FakeEngine, quiet=True)
    root = str(tmp_path / 'many')
    mp.spawn(_rank, args=(world, _free_port(), root, 'plain', str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f'rank{r}.txt').read() == 'ok'
    for li, l1 in enumerate(one):
        for rel in OUTPUTS:
            assert filecmp.cmp(os.path.join(l1.path, rel), os.path.join(root, f'locus{li}', rel), shallow=False), (li, rel)


def test_more_ranks_than_reads_leaves_empty_shards(tmp_path):
    """Four reads on eight ranks: four ranks own nothing and still take part in both collectives."""
    global LOCI
    saved = LOCI
    try:
        LOCI = [('(AGC)', 16, 3), ('(AAAT)', 30, 1)]
        one, sig = _make_loci(str(tmp_path / 'one'))
        main_wrapper_loci(one, 1, signal_loader=_loader(sig), _engine=FakeEngine, quiet=True)
    finally:
        LOCI = saved
    root = str(tmp_path / 'many')
    mp.spawn(_rank_few, args=(8, _free_port(), root, str(tmp_path)), nprocs=8, join=True)
    for r in range(8):
        assert open(tmp_path / f'rank{r}.txt').read() == 'ok'
    for li, l1 in enumerate(one):
        for rel in OUTPUTS:
            assert filecmp.cmp(os.path.join(l1.path, rel), os.path.join(root, f'locus{li}', rel), shallow=False), (li, rel)


def _rank_few(rank, world, port, root, out_dir):
    global LOCI
    LOCI = [('(AGC)', 16, 3), ('(AAAT)', 30, 1)]
    _rank(rank, world, port, root, 'plain', out_dir)


def test_a_failing_read_raises_on_every_rank_alike(tmp_path):
    """A read the caller fails on (status != 0) travels through the collectives like any other; every rank then raises
    upstream's error for it -- nobody is left waiting -- and the loci before it are complete."""
    mp.spawn(_rank, args=(2, _free_port(), str(tmp_path / 'many'), 'poison', str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        msg = open(tmp_path / f'rank{r}.txt').read()
        assert msg.startswith('ReadCallError') and 'L1_read004' in msg and 'fit_points' in msg
    assert os.path.exists(tmp_path / 'many' / 'locus0' / 'predictions' / 'sequences' / 'all.fasta')
    assert not os.path.exists(tmp_path / 'many' / 'locus1' / 'predictions' / 'sequences' / 'all.fasta')


def test_an_exception_on_one_rank_reaches_all(tmp_path):
    """A missing file on ONE rank (before the collectives): the rank raises its own error, the others a RuntimeError naming
    it, instead of waiting in the all-gather until the backend times out."""
    world = 4
    mp.spawn(_rank, args=(world, _free_port(), str(tmp_path / 'many'), 'missing', str(tmp_path)), nprocs=world, join=True)
    msgs = [open(tmp_path / f'rank{r}.txt').read() for r in range(world)]
    assert sum(m.startswith('FileNotFoundError') for m in msgs) == 1
    assert sum(m.startswith('RuntimeError') and 'failed on rank' in m and 'is missing' in m for m in msgs) == world - 1


def test_one_rank_failing_read_raises(tmp_path):
    loci, sig = _make_loci(str(tmp_path / 'a'), poison='L0_read001')
    with pytest.raises(ReadCallError, match='L0_read001'):
        main_wrapper_loci(loci, 1, signal_loader=_loader(sig), _engine=FakeEngine, quiet=True)


def test_threads_for_the_per_locus_host_work(tmp_path):
    """threads > 1: overviews / automata / output files on threads of this process (native code without the GIL) -- the same
    files as one thread writes."""
    global LOCI
    saved = LOCI
    try:
        LOCI = [(p, fl, 2 + (i % 3)) for i, (p, fl) in enumerate([('(AGC)', 16), ('(AAAT)', 30), ('(CAG)CAACAG(CCG)', 20), ('(GGCCCC)', 24)] * 17)]
        a, sig = _make_loci(str(tmp_path / 'a'))
        b, _ = _make_loci(str(tmp_path / 'b'))
    finally:
        LOCI = saved
    tm = {}
    main_wrapper_loci(a, 3, signal_loader=_loader(sig), _engine=FakeEngine, quiet=True, timings=tm)
    main_wrapper_loci(b, 1, signal_loader=_loader(sig), _engine=FakeEngine, quiet=True)
    assert tm['host_threads'] == 3 and tm['n_loci'] == 68 and tm['reader_processes'] == 0
    for la, lb in zip(a, b):
        for rel in OUTPUTS:
            assert filecmp.cmp(os.path.join(la.path, rel), os.path.join(lb.path, rel), shallow=False), rel


def test_raw_reads_and_the_byte_budget_of_a_batch(tmp_path):
    """The default path hands whole raw int16 reads to the engine; a batch is closed early when its raw bytes exceed the budget
    (long reads): the cut does not show in the outputs."""
    a, sig = _make_loci(str(tmp_path / 'a'))
    b, _ = _make_loci(str(tmp_path / 'b'))
    rng = np.random.default_rng(9)
    raws = {}
    for nm, s in sig.items():  # the segment sits at [1000, 1000 + len - 1] of a longer read, as the overviews say
        raws[nm] = np.concatenate([rng.integers(400, 600, 1000), np.round(s * 64).astype(np.int64) + 500, rng.integers(400, 600, 300)]).astype(np.int16)
    reader = lambda path: raws[os.path.basename(path)[:-len('.fast5')]]
    tm_a, tm_b = {}, {}
    main_wrapper_loci(a, 1, raw_reader=reader, _engine=FakeEngine, quiet=True, batch_raw_bytes=40000, timings=tm_a)
    main_wrapper_loci(b, 1, raw_reader=reader, _engine=FakeEngine, quiet=True, timings=tm_b)
    assert tm_a['batches'] > tm_b['batches'] == 1
    for la, lb in zip(a, b):
        for rel in OUTPUTS:
            assert filecmp.cmp(os.path.join(la.path, rel), os.path.join(lb.path, rel), shallow=False), rel


def test_fast5_files_are_read_on_the_worker_processes(tmp_path):
    """From 64 loci on with threads > 1 the batches' fast5 files are opened and decoded on the worker processes (the upstream
    test file's ten VBZ-compressed reads, 70 loci pointing at them through the caller-only `fast5_path` column): the same files as
    one process writes."""
    from tests.helpers import GOLDEN
    from warpstr_amd import fast5
    try:
        fast5._libs()
    except fast5.Fast5Error as e:
        pytest.skip(str(e))
    src = os.path.join(GOLDEN, 'real', 'batch_0.fast5')
    ids = fast5.Fast5File(src).read_ids()[:10]
    pats = [('(AGC)', 16), ('(AAAT)', 30), ('(CAG)CAACAG(CCG)', 20), ('(GGCCCC)', 24)]

    def make(root):
        loci = []
        for li in range(70):
            pattern, fl = pats[li % 4]
            locus = synth.make_locus(pattern, fl, 900 + li)
            loc = os.path.join(root, f'locus{li}')
            ov.store_flanks(loc, [locus.left_t, locus.right_t, locus.left_r, locus.right_r])
            rows = [ids[(li + k) % 10] for k in range(1 + li % 3)]
            pd.DataFrame({'read_name': rows, 'run_id': 'run_0', 'reverse': [bool((li + k) & 1) for k in range(len(rows))], 'saved': 1,
                          'l_start_raw': 5000 + 10 * li, 'r_end_raw': 6500 + 10 * li, 'fast5_path': src}).to_csv(os.path.join(loc, 'overview.csv'), index=False)
            loci.append(LocusPath(loc, pattern, fl))
        return loci
    a, b, c = make(str(tmp_path / 'a')), make(str(tmp_path / 'b')), make(str(tmp_path / 'c'))
    tm, tm_c = {}, {}
    main_wrapper_loci(a, 3, _engine=FakeEngine, quiet=True, timings=tm)
    main_wrapper_loci(b, 1, _engine=FakeEngine, quiet=True)
    assert tm['reader_processes'] == 3 and tm['n_reads'] == sum(1 + li % 3 for li in range(70)) >= 64
    # ... and decoded by the workers straight into the staging buffers both sides map (lengths first, then every read to its
    # place), the batches cut by the raw-byte budget
    main_wrapper_loci(c, 3, _engine=SharedFakeEngine, quiet=True, timings=tm_c, batch_raw_bytes=12 << 20)
    assert 3 <= tm_c['shared_batches'] <= tm_c['batches'] and tm_c['reader_processes'] == 3   # (the tail of fewer than 64 reads is read here)
    # ... and into arenas of their own, handed out a batch ahead (batches of 40 reads here: several generations per region)
    import warpstr_amd.loci as wl
    d, tm_d = make(str(tmp_path / 'd')), {}
    old_reads = wl.SHARED_BATCH_READS
    wl.SHARED_BATCH_READS = 40
    try:
        main_wrapper_loci(d, 3, _engine=ArenaFakeEngine, quiet=True, timings=tm_d)
    finally:
        wl.SHARED_BATCH_READS = old_reads
    assert tm_d['reader_mode'] == 'arenas' and tm_d['arena_batches'] == tm_d['batches'] >= 4 and tm_d['shared_batches'] == 0
    # ... and, for an engine that decodes VBZ itself, as the StreamVByte blocks inside the chunks' zstd frames
    e, tm_e = make(str(tmp_path / 'e')), {}
    wl.SHARED_BATCH_READS = 40
    try:
        main_wrapper_loci(e, 3, _engine=VbzFakeEngine, quiet=True, timings=tm_e)
    finally:
        wl.SHARED_BATCH_READS = old_reads
    assert tm_e['reader_mode'] == 'arenas, VBZ decoded on the GPU' and tm_e['vbz_batches'] == tm_e['batches'] >= 4 and tm_e['arena_batches'] == 0
    # (without _host_loci.so the readers decode with NumPy and hand plain samples over: the same files, no fewer bytes)
    assert tm_e['uploaded_bytes'] < (0.75 if fast5._vbz_native() is not None else 1.01) * tm_e['raw_bytes'] and tm_d['uploaded_bytes'] == tm_d['raw_bytes']
    # ... and a region is not handed out again before the calling thread has SUBMITTED the batch that used it (an engine that
    # dawdles before it looks at the arenas: the readers would have overwritten them when taking a batch from the queue was enough)
    import time

    class Dawdling(ArenaFakeEngine):
        def submit_raw_parts(self, *args):
            time.sleep(0.05)
            return super().submit_raw_parts(*args)
    f, tm_f = make(str(tmp_path / 'f')), {}
    wl.SHARED_BATCH_READS = 16
    try:
        main_wrapper_loci(f, 3, _engine=Dawdling, quiet=True, timings=tm_f)
    finally:
        wl.SHARED_BATCH_READS = old_reads
    assert tm_f['arena_batches'] >= 8
    for lf, lb in zip(f, b):
        for rel in OUTPUTS:
            assert filecmp.cmp(os.path.join(lf.path, rel), os.path.join(lb.path, rel), shallow=False), rel
    # ... and without reader processes the same arenas are filled by the reader thread of this process
    for tag, engine in (('g', ArenaFakeEngine), ('h', VbzFakeEngine)):
        g, tm_g = make(str(tmp_path / tag)), {}
        wl.SHARED_BATCH_READS = 64   # (an inline batch is a quarter of it)
        try:
            main_wrapper_loci(g, 1, _engine=engine, quiet=True, timings=tm_g)
        finally:
            wl.SHARED_BATCH_READS = old_reads
        assert tm_g['reader_mode'].endswith('filled in this process') and tm_g['reader_processes'] == 0 and tm_g['batches'] >= 8
        assert tm_g['vbz_batches' if engine is VbzFakeEngine else 'arena_batches'] == tm_g['batches']
        for lg, lb in zip(g, b):
            for rel in OUTPUTS:
                assert filecmp.cmp(os.path.join(lg.path, rel), os.path.join(lb.path, rel), shallow=False), rel
    for la, lb, lc, ld, le in zip(a, b, c, d, e):
        for rel in OUTPUTS:
            assert filecmp.cmp(os.path.join(la.path, rel), os.path.join(lb.path, rel), shallow=False), rel
            assert filecmp.cmp(os.path.join(lc.path, rel), os.path.join(lb.path, rel), shallow=False), rel
            assert filecmp.cmp(os.path.join(ld.path, rel), os.path.join(lb.path, rel), shallow=False), rel
            assert filecmp.cmp(os.path.join(le.path, rel), os.path.join(lb.path, rel), shallow=False), rel
    assert not [f for f in os.listdir('/dev/shm') if f.startswith('warpstr_arena_')]
    assert not [f for f in os.listdir('/dev/shm') if f.startswith('warpstr_stage_')]


def test_native_and_pandas_host_paths_write_the_same_files(tmp_path):
    """The native per-locus host work (csrc/host_loci.cpp) against the pandas / Python forms it replaces: every output file
    byte for byte, the returned tables equal; and a table the native parser declines (a quoted field) takes the pandas path
    inside a native run."""
    a, sig = _make_loci(str(tmp_path / 'a'))
    b, _ = _make_loci(str(tmp_path / 'b'))
    for loci in (a, b):   # one table the native parser will not touch
        p = os.path.join(loci[2].path, 'overview.csv')
        df = pd.read_csv(p)
        df['note'] = ['with, comma'] + ['x'] * (len(df) - 1)
        df.to_csv(p, index=False)
    tm_a, tm_b = {}, {}
    ta = main_wrapper_loci(a, 1, signal_loader=_loader(sig), _engine=FakeEngine, quiet=True, timings=tm_a)
    tb = main_wrapper_loci(b, 1, signal_loader=_loader(sig), _engine=FakeEngine, quiet=True, native=False, timings=tm_b)
    from warpstr_amd import _hostlib
    if _hostlib.lib() is not None:
        assert tm_a['native_overviews'] == len(a) - 1
    assert tm_b['native_overviews'] == 0
    for la, lb, (dfa, ca), (dfb, cb) in zip(a, b, ta, tb):
        for rel in OUTPUTS:
            assert filecmp.cmp(os.path.join(la.path, rel), os.path.join(lb.path, rel), shallow=False), rel
        pd.testing.assert_frame_equal(dfa, dfb)
        assert (ca is None) == (cb is None)
        if ca is not None:
            pd.testing.assert_frame_equal(ca, cb)
            rel = 'predictions/complexSTR_analysis/complex_repeat_units.csv'
            assert filecmp.cmp(os.path.join(la.path, rel), os.path.join(lb.path, rel), shallow=False)
    assert any(c is not None for _, c in ta)


def test_reads_in_host_memory_by_name(tmp_path):
    """raw_reads={name: int16 read}: the same outputs as a raw_reader call-back per read."""
    a, sig = _make_loci(str(tmp_path / 'a'))
    b, _ = _make_loci(str(tmp_path / 'b'))
    rng = np.random.default_rng(9)
    raws = {nm: np.concatenate([rng.integers(400, 600, 1000), np.round(s * 64).astype(np.int64) + 500]).astype(np.int16) for nm, s in sig.items()}
    main_wrapper_loci(a, 2, raw_reads=raws, _engine=FakeEngine, quiet=True, batch_reads=7)
    main_wrapper_loci(b, 1, raw_reader=lambda path: raws[os.path.basename(path)[:-len('.fast5')]], _engine=FakeEngine, quiet=True)
    for la, lb in zip(a, b):
        for rel in OUTPUTS:
            assert filecmp.cmp(os.path.join(la.path, rel), os.path.join(lb.path, rel), shallow=False), rel


def test_store_time_does_not_grow_with_the_run(tmp_path):
    """The outputs of a locus are made from ITS slice of the run's sequence buffers (each locus once decoded the whole run's
    buffers: quadratic).  400 loci through the pandas path: the store phase stays a small multiple of 100 loci's."""
    global LOCI
    saved = LOCI
    try:
        LOCI = [('(AGC)', 16, 3)] * 400
        a, sig = _make_loci(str(tmp_path / 'a'))
    finally:
        LOCI = saved
    tm_small, tm_big = {}, {}
    main_wrapper_loci(a[:100], 1, signal_loader=_loader(sig), _engine=FakeEngine, quiet=True, native=False, timings=tm_small)
    main_wrapper_loci(a, 1, signal_loader=_loader(sig), _engine=FakeEngine, quiet=True, native=False, timings=tm_big)
    assert tm_big['store_s'] < 8 * tm_small['store_s'] + 0.5, (tm_big['store_s'], tm_small['store_s'])


def _rank_many(rank, world, port, root, out_dir, mode):
    import json

    import torch.distributed as dist
    global LOCI
    LOCI = [(p, fl, 1 + (i * 7) % 5) for i, (p, fl) in enumerate([('(AGC)', 16), ('(AAAT)', 30), ('(CAG)CAACAG(CCG)', 20), ('(GGCCCC)', 24)] * 18)]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    loci, sig = _make_loci(root) if rank == 0 else (None, None)
    dist.barrier()
    if rank != 0:
        import tempfile
        with tempfile.TemporaryDirectory() as scratch:
            _, sig = _make_loci(scratch)
        loci = [LocusPath(os.path.join(root, f'locus{li}'), p, fl) for li, (p, fl, _) in enumerate(LOCI)]
    if mode == 'poison':
        sig['L40_read000'] = sig['L40_read000'].copy()
        sig['L40_read000'][0] = 99.0
    tm, msg, lens = {}, 'ok', None
    try:
        tables = main_wrapper_loci(loci, 2, signal_loader=_loader(sig), _engine=FakeEngine, quiet=True, shard=True, timings=tm)
        lens = [int((np.asarray(tables[i][0]['results']) >= 0).sum()) for i in (0, 35, 71)]   # other ranks' loci come from their files
    except Exception as e:  # noqa: BLE001
        msg = f'{type(e).__name__}: {e}'
    with open(os.path.join(out_dir, f'rank{rank}.json'), 'w') as f:
        json.dump({'msg': msg, 'set_up': tm.get('loci_set_up'), 'partition': tm.get('partition'), 'lens': lens}, f)
    dist.destroy_process_group()


def test_many_loci_are_partitioned_by_locus_over_the_ranks(tmp_path):
    """72 loci on 8 ranks (>= 8 per rank): every rank sets up, calls and writes ONLY its own loci -- together every locus exactly
    once --, no result travels, and the files equal one rank's."""
    import json
    global LOCI
    saved = LOCI
    try:
        LOCI = [(p, fl, 1 + (i * 7) % 5) for i, (p, fl) in enumerate([('(AGC)', 16), ('(AAAT)', 30), ('(CAG)CAACAG(CCG)', 20), ('(GGCCCC)', 24)] * 18)]
        one, sig = _make_loci(str(tmp_path / 'one'))
        n_reads = [n for _, _, n in LOCI]
    finally:
        LOCI = saved
    main_wrapper_loci(one, 1, signal_loader=_loader(sig), _engine=FakeEngine, quiet=True)
    world, root = 8, str(tmp_path / 'many')
    mp.spawn(_rank_many, args=(world, _free_port(), root, str(tmp_path), 'plain'), nprocs=world, join=True)
    got = [json.load(open(tmp_path / f'rank{r}.json')) for r in range(world)]
    assert all(g['msg'] == 'ok' and g['partition'] == 'loci' for g in got), got
    owned = sorted(i for g in got for i in g['set_up'])
    assert owned == list(range(72))                       # every locus on exactly one rank
    assert all(4 <= len(g['set_up']) <= 14 for g in got)   # ... and the ranks share them
    assert all(g['lens'] == [n_reads[0], n_reads[35], n_reads[71]] for g in got)
    for li, l1 in enumerate(one):
        for rel in OUTPUTS:
            assert filecmp.cmp(os.path.join(l1.path, rel), os.path.join(root, f'locus{li}', rel), shallow=False), (li, rel)


def test_locus_partition_a_failing_read_stops_every_rank_where_upstream_would_stop(tmp_path):
    """Partition by locus, a read of locus 40 fails on the rank that owns it: that rank raises upstream's error, the others an
    error naming it; the loci before 40 are complete on every rank, locus 40 and later ones are not written."""
    import json
    world, root = 4, str(tmp_path / 'many')
    mp.spawn(_rank_many, args=(world, _free_port(), root, str(tmp_path), 'poison'), nprocs=world, join=True)
    msgs = [json.load(open(tmp_path / f'rank{r}.json'))['msg'] for r in range(world)]
    assert sum(m.startswith('ReadCallError') and 'L40_read000' in m for m in msgs) == 1, msgs
    assert sum(m.startswith('RuntimeError') and 'L40_read000' in m for m in msgs) == world - 1, msgs
    fasta = lambda li: os.path.exists(os.path.join(root, f'locus{li}', 'predictions', 'sequences', 'all.fasta'))
    assert all(fasta(li) for li in range(40)) and not any(fasta(li) for li in range(40, 72))


def _fast5_loci(root, src, ids, n_loci=70, missing=None):
    loci = []
    pats = [('(AGC)', 16), ('(AAAT)', 30), ('(CAG)CAACAG(CCG)', 20), ('(GGCCCC)', 24)]
    for li in range(n_loci):
        pattern, fl = pats[li % 4]
        locus = synth.make_locus(pattern, fl, 900 + li)
        loc = os.path.join(root, f'locus{li}')
        ov.store_flanks(loc, [locus.left_t, locus.right_t, locus.left_r, locus.right_r])
        rows = [ids[(li + k) % 10] for k in range(1 + li % 3)]
        pd.DataFrame({'read_name': rows, 'run_id': 'run_0', 'reverse': [bool((li + k) & 1) for k in range(len(rows))], 'saved': 1,
                      'l_start_raw': 5000 + 10 * li, 'r_end_raw': 6500 + 10 * li,
                      'fast5_path': src if li != missing else src + '.gone'}).to_csv(os.path.join(loc, 'overview.csv'), index=False)
        loci.append(LocusPath(loc, pattern, fl))
    return loci


def test_reader_processes_that_cannot_start_are_reported_and_what_a_worker_raises_is_raised(tmp_path, monkeypatch, capsys):
    """Only the START of the reader processes falls back to reading in this process (said once on stderr, same files); an error
    raised INSIDE a worker -- a fast5 file that is not there -- is raised here, not turned into a slower run."""
    import sys
    from tests.helpers import GOLDEN
    from warpstr_amd import fast5
    try:
        fast5._libs()
    except fast5.Fast5Error as e:
        pytest.skip(str(e))
    src = os.path.join(GOLDEN, 'real', 'batch_0.fast5')
    ids = fast5.Fast5File(src).read_ids()[:10]
    a, b = _fast5_loci(str(tmp_path / 'a'), src, ids), _fast5_loci(str(tmp_path / 'b'), src, ids)
    main_wrapper_loci(b, 1, _engine=FakeEngine, quiet=True)
    with monkeypatch.context() as mp_:
        mp_.setattr(sys, 'executable', str(tmp_path / 'no-such-python'))
        tm = {}
        main_wrapper_loci(a, 3, _engine=FakeEngine, quiet=True, timings=tm)
    assert tm['reader_processes'] == 0 and 'could not start 3 reader processes' in capsys.readouterr().err
    for la, lb in zip(a, b):
        for rel in OUTPUTS:
            assert filecmp.cmp(os.path.join(la.path, rel), os.path.join(lb.path, rel), shallow=False), rel
    c = _fast5_loci(str(tmp_path / 'c'), src, ids, missing=40)
    with pytest.raises(RuntimeError, match='failed in a worker process'):
        main_wrapper_loci(c, 3, _engine=SharedFakeEngine, quiet=True)


def test_garbage_collector_paused_during_a_long_run_and_back_afterwards(tmp_path, monkeypatch):
    """main_wrapper_loci pauses the cyclic collector for a run of GC_PAUSE_FROM_LOCI loci or more -- and only then, only if it was
    on, and it is on again after the call, also when the call raises."""
    import gc
    seen = []

    class Probe(FakeEngine):
        def __init__(self, *a, **k):
            seen.append(gc.isenabled())
            super().__init__(*a, **k)

    class Failing(FakeEngine):
        def __init__(self, *a, **k):
            seen.append(gc.isenabled())
            raise RuntimeError('no handle')
    from warpstr_amd import loci as loci_mod
    monkeypatch.delenv('WARPSTR_KEEP_GC', raising=False)
    monkeypatch.setattr(loci_mod, 'GC_PAUSE_FROM_LOCI', 3)
    loci, sig = _make_loci(str(tmp_path / 'a'))
    kw = dict(signal_loader=_loader(sig), quiet=True)
    assert gc.isenabled()
    loci_mod.main_wrapper_loci(loci, 1, **kw, _engine=Probe)
    assert seen == [False] and gc.isenabled()
    with pytest.raises(RuntimeError, match='no handle'):
        loci_mod.main_wrapper_loci(loci, 1, **kw, _engine=Failing)
    assert seen == [False, False] and gc.isenabled()
    loci_mod.main_wrapper_loci(loci[:2], 1, **kw, _engine=Probe)   # a short run: left alone
    assert seen[-1] is True
    monkeypatch.setenv('WARPSTR_KEEP_GC', '1')
    loci_mod.main_wrapper_loci(loci, 1, **kw, _engine=Probe)
    assert seen[-1] is True
    monkeypatch.delenv('WARPSTR_KEEP_GC')
    gc.disable()
    try:
        loci_mod.main_wrapper_loci(loci, 1, **kw, _engine=Probe)
        assert not gc.isenabled()   # it was off before the call: it stays off
    finally:
        gc.enable()


def test_readers_decode_while_the_handle_is_created_and_stop_if_it_cannot_be(tmp_path):
    """With arenas the reader thread is started BEFORE the engine: its constructor already finds the thread running (and, given
    time, the first chunks decoded); an engine that cannot be created ends the run with its own error, the thread and the reader
    processes gone; a worker's error is still raised (a fast5 file that is not there)."""
    import threading
    import time
    from tests.helpers import GOLDEN
    from warpstr_amd import fast5
    try:
        fast5._libs()
    except fast5.Fast5Error as e:
        pytest.skip(str(e))
    if not os.path.isdir('/dev/shm'):
        pytest.skip('no /dev/shm')
    src = os.path.join(GOLDEN, 'real', 'batch_0.fast5')
    ids = fast5.Fast5File(src).read_ids()[:10]
    seen = {}

    class Slow(ArenaFakeEngine):
        def __init__(self, *a):
            seen['reader_running'] = any(t.name == 'warpstr-reader' for t in threading.enumerate())
            time.sleep(0.5)
            seen['arena_files'] = len([f for f in os.listdir('/dev/shm') if f.startswith('warpstr_arena_')])
            super().__init__(*a)

    class Broken(ArenaFakeEngine):
        def __init__(self, *a):
            raise RuntimeError('no device')
    a, b = _fast5_loci(str(tmp_path / 'a'), src, ids), _fast5_loci(str(tmp_path / 'b'), src, ids)
    main_wrapper_loci(b, 1, _engine=FakeEngine, quiet=True)
    tm = {}
    main_wrapper_loci(a, 3, _engine=Slow, quiet=True, timings=tm)
    assert seen['reader_running'] and seen['arena_files'] >= 1 and tm['reader_mode'] == 'arenas' and tm['arena_batches'] >= 1
    for la, lb in zip(a, b):
        for rel in OUTPUTS:
            assert filecmp.cmp(os.path.join(la.path, rel), os.path.join(lb.path, rel), shallow=False), rel
    c = _fast5_loci(str(tmp_path / 'c'), src, ids)
    with pytest.raises(RuntimeError, match='no device'):
        main_wrapper_loci(c, 3, _engine=Broken, quiet=True)
    assert not any(t.name == 'warpstr-reader' for t in threading.enumerate())
    d = _fast5_loci(str(tmp_path / 'd'), src, ids, missing=40)
    with pytest.raises(RuntimeError, match='failed in a worker process'):
        main_wrapper_loci(d, 3, _engine=ArenaFakeEngine, quiet=True)
    assert not any(t.name == 'warpstr-reader' for t in threading.enumerate())


def test_a_reader_process_that_dies_ends_the_run_with_an_error(tmp_path):
    """A reader process killed in the middle of a run: the chunk it held fails (its pipe ends), the run raises -- it neither hangs
    nor writes files from half a batch -- and the other reader processes are gone afterwards."""
    import threading
    import psutil
    from tests.helpers import GOLDEN
    from warpstr_amd import fast5
    import warpstr_amd.loci as wl
    try:
        fast5._libs()
    except fast5.Fast5Error as e:
        pytest.skip(str(e))
    src = os.path.join(GOLDEN, 'real', 'batch_0.fast5')
    ids = fast5.Fast5File(src).read_ids()[:10]
    me = psutil.Process()
    before = {p.pid for p in me.children(recursive=True)}
    killed = []

    class Killer(ArenaFakeEngine):
        def submit_raw_parts(self, *args):
            if not killed:
                workers = [p for p in me.children(recursive=True) if p.pid not in before and '_hostworker' in ' '.join(p.cmdline())]
                if len(workers) == 4:   # (the readers were forked by one helper process: they are its children)
                    workers = [p for p in workers if p.ppid() != me.pid]
                assert len(workers) == 3
                workers[0].kill()
                killed.append(workers[0].pid)
            return super().submit_raw_parts(*args)
    loci = _fast5_loci(str(tmp_path / 'a'), src, ids, n_loci=200)
    old = wl.SHARED_BATCH_READS
    wl.SHARED_BATCH_READS = 16
    done = []

    def run():
        try:
            main_wrapper_loci(loci, 3, _engine=Killer, quiet=True)
            done.append(None)
        except BaseException as e:  # noqa: BLE001
            done.append(e)
    try:
        t = threading.Thread(target=run, daemon=True)
        t.start()
        t.join(60)
        assert not t.is_alive(), 'the run hangs after a reader process died'
    finally:
        wl.SHARED_BATCH_READS = old
    assert killed and isinstance(done[0], (EOFError, OSError, RuntimeError)), done
    assert not [p for p in me.children(recursive=True) if p.pid not in before and p.is_running() and p.status() != psutil.STATUS_ZOMBIE]
    assert not os.path.exists(os.path.join(loci[-1].path, 'predictions', 'sequences', 'all.fasta'))
    assert not [f for f in os.listdir('/dev/shm') if f.startswith(f'warpstr_arena_{killed[0]}_')]   # (the dead reader's arenas too)


def test_arenas_are_not_used_when_dev_shm_has_no_room_for_them(tmp_path, monkeypatch):
    """A container's default /dev/shm (64 MB) cannot hold the reader arenas: the run says so in its timings and reads through the
    staging ring (reader processes) or the page-locked ring (one process) instead -- same files."""
    from tests.helpers import GOLDEN
    from warpstr_amd import fast5
    import warpstr_amd.loci as wl
    try:
        fast5._libs()
    except fast5.Fast5Error as e:
        pytest.skip(str(e))
    src = os.path.join(GOLDEN, 'real', 'batch_0.fast5')
    ids = fast5.Fast5File(src).read_ids()[:10]
    a, b, c = (_fast5_loci(str(tmp_path / t), src, ids) for t in 'abc')
    main_wrapper_loci(b, 1, _engine=FakeEngine, quiet=True)
    monkeypatch.setattr(wl, 'ARENA_ROOM_PER_READ', 1 << 40)   # (as if every read took a terabyte -- and the batches' byte budget let it)
    tm_a, tm_c = {}, {}
    main_wrapper_loci(a, 3, _engine=ArenaFakeEngine, quiet=True, timings=tm_a, batch_raw_bytes=1 << 44)
    main_wrapper_loci(c, 1, _engine=ArenaFakeEngine, quiet=True, timings=tm_c, batch_raw_bytes=1 << 44)
    assert 'MB free' in tm_a['arenas_refused'] and tm_a['reader_mode'] == 'shared staging' and tm_a['arena_batches'] == 0 and tm_a['shared_batches'] > 0
    assert 'MB free' in tm_c['arenas_refused'] and tm_c['arena_batches'] == 0 and tm_c['local_batches'] > 0
    for la, lb, lc in zip(a, b, c):
        for rel in OUTPUTS:
            assert filecmp.cmp(os.path.join(la.path, rel), os.path.join(lb.path, rel), shallow=False), rel
            assert filecmp.cmp(os.path.join(lc.path, rel), os.path.join(lb.path, rel), shallow=False), rel


def test_a_few_loci_with_many_reads_get_reader_processes_too(tmp_path, monkeypatch):
    """configs[3] / [4]: a handful of loci with thousands of reads each.  The reader processes are started by the size of the run's
    overviews, not only by the number of loci: three loci x 50 reads with the threshold lowered to 100 reads."""
    import warpstr_amd.loci as wl
    from tests.helpers import GOLDEN
    from warpstr_amd import fast5
    try:
        fast5._libs()
    except fast5.Fast5Error as e:
        pytest.skip(str(e))
    src = os.path.join(GOLDEN, 'real', 'batch_0.fast5')
    ids = fast5.Fast5File(src).read_ids()[:10]

    def make(root):
        loci = []
        for li, (pattern, fl) in enumerate([('(AGC)', 16), ('(AAAT)', 30), ('(GGCCCC)', 24)]):
            locus = synth.make_locus(pattern, fl, 900 + li)
            loc = os.path.join(root, f'locus{li}')
            ov.store_flanks(loc, [locus.left_t, locus.right_t, locus.left_r, locus.right_r])
            rows = [ids[(li + k) % 10] for k in range(50)]
            pd.DataFrame({'read_name': rows, 'run_id': 'run_0', 'reverse': [bool(k & 1) for k in range(50)], 'saved': 1, 'l_start_raw': 5000,
                          'r_end_raw': 6500, 'fast5_path': src}).to_csv(os.path.join(loc, 'overview.csv'), index=False)
            loci.append(LocusPath(loc, pattern, fl))
        return loci
    a, b = make(str(tmp_path / 'a')), make(str(tmp_path / 'b'))
    tm_a, tm_b = {}, {}
    main_wrapper_loci(a, 3, _engine=VbzFakeEngine, quiet=True, timings=tm_a)     # default thresholds: 150 reads are a small run
    assert tm_a['reader_processes'] == 0 and tm_a['reader_mode'].endswith('filled in this process')
    monkeypatch.setattr(wl, 'READER_POOL_FROM_READS', 100)
    main_wrapper_loci(b, 3, _engine=VbzFakeEngine, quiet=True, timings=tm_b)
    assert tm_b['reader_processes'] == 3 and tm_b['reader_mode'] == 'arenas, VBZ decoded on the GPU'
    for la, lb in zip(a, b):
        for rel in OUTPUTS:
            assert filecmp.cmp(os.path.join(la.path, rel), os.path.join(lb.path, rel), shallow=False), rel


def test_arena_batches_are_cut_at_the_byte_budget(tmp_path):
    """Reader arenas: a batch holds at most batch_raw_bytes / 2 of raw samples, counted -- before anything of a read is known --
    as the read's segment end (2 (r_end_raw + 1) bytes at least) or the mean of the reads decoded so far, whichever is more.
    The test file's reads are 59-170 k samples: with a budget of 1 MB a batch holds a handful of reads instead of all of them;
    same files as the run without a budget to speak of."""
    from tests.helpers import GOLDEN
    from warpstr_amd import fast5
    try:
        fast5._libs()
    except fast5.Fast5Error as e:
        pytest.skip(str(e))
    src = os.path.join(GOLDEN, 'real', 'batch_0.fast5')
    ids = fast5.Fast5File(src).read_ids()[:10]
    sizes = []

    class Counting(ArenaFakeEngine):
        def submit_raw_parts(self, region, parts, lo, hi, aut):
            sizes.append(2 * int(sum(sum(p[3]) for p in parts)))
            return super().submit_raw_parts(region, parts, lo, hi, aut)
    a, b = (_fast5_loci(str(tmp_path / t), src, ids, n_loci=80) for t in 'ab')
    main_wrapper_loci(b, 1, _engine=FakeEngine, quiet=True)
    tm = {}
    main_wrapper_loci(a, 3, _engine=Counting, quiet=True, timings=tm, batch_raw_bytes=2 << 20)
    assert tm['arena_batches'] == len(sizes) > 8
    # the batches handed out before anything was decoded know the segment ends only (13-15 kB a read, 64 reads at most); from
    # then on the mean of what was decoded counts
    assert np.median(sizes[2:]) <= (1 << 20) and max(sizes[2:]) <= (2 << 20), sizes   # (counted by the mean: a batch of long reads runs over)
    assert max(sizes[:2]) <= 64 * 400_000
    for la, lb in zip(a, b):
        for rel in OUTPUTS:
            assert filecmp.cmp(os.path.join(la.path, rel), os.path.join(lb.path, rel), shallow=False), rel


def test_a_batch_whose_arena_finds_no_room_is_read_through_the_pipes(tmp_path, monkeypatch, capsys):
    """/dev/shm fills up under a run (or its reads are far longer than the room check assumed): the reader's OSError ends the
    arena path for THAT batch only -- its reads come back through the pipes and go up from the page-locked ring, said once on
    stderr -- and the run's files are the same."""
    from tests.helpers import GOLDEN
    from warpstr_amd import fast5
    import warpstr_amd.loci as wl
    try:
        fast5._libs()
    except fast5.Fast5Error as e:
        pytest.skip(str(e))
    src = os.path.join(GOLDEN, 'real', 'batch_0.fast5')
    ids = fast5.Fast5File(src).read_ids()[:10]
    a, b = (_fast5_loci(str(tmp_path / t), src, ids, n_loci=70) for t in 'ab')
    main_wrapper_loci(b, 1, _engine=FakeEngine, quiet=True)
    calls = []
    real = wl._decode_arena

    def refusing(args):   # (in-process readers: the arena function of every second batch finds /dev/shm full)
        calls.append(args[1])
        if args[1] % 2 == 1:
            raise OSError('/dev/shm has no room for a reader arena of 123 bytes (WARPSTR_NO_READER_ARENAS=1 reads without arenas)')
        return real(args)
    refusing.__name__ = '_decode_arena'
    monkeypatch.setattr(wl, '_decode_arena', refusing)
    monkeypatch.setattr(wl, 'SHARED_BATCH_READS', 64)   # (an in-process run cuts a quarter of it: batches of 16 reads)
    tm = {}

    class Inline(wl._InlinePool):
        def submit(self, func, item):
            from concurrent.futures import Future
            if func is refusing:
                future = Future()
                try:
                    future.set_result(func((f'{self.owner}{item[0]}',) + tuple(item[1:])))
                except OSError as e:   # (what a worker process reports: a RuntimeError carrying the traceback text)
                    future.set_exception(RuntimeError(f'_decode_arena failed in a worker process:\nOSError: {e}'))
                return future
            return super().submit(func, item)
    monkeypatch.setattr(wl, '_InlinePool', Inline)
    main_wrapper_loci(a, 1, _engine=ArenaFakeEngine, quiet=True, timings=tm)
    assert tm['arena_fallbacks'] >= 2 and tm['arena_batches'] >= 2 and len(set(calls)) >= 4, tm
    err = capsys.readouterr().err
    assert err.count('such batches are read without arenas') == 1
    for la, lb in zip(a, b):
        for rel in OUTPUTS:
            assert filecmp.cmp(os.path.join(la.path, rel), os.path.join(lb.path, rel), shallow=False), rel


@pytest.mark.parametrize('threads', [1, 3])
def test_a_long_run_streams_its_set_up_reading_and_calling(tmp_path, threads, monkeypatch):
    """From STREAM_FROM_LOCI loci on (one rank, fast5 files, reader arenas) the run is one pipeline: the loci are set up part after
    part on a thread of its own, the readers start on the first part's files, the handle is created from the loci known by then
    and takes the later ones with add_automata while batches are in flight.  Same files as the run that sets everything up first;
    the timings say which path ran."""
    from tests.helpers import GOLDEN
    from warpstr_amd import fast5
    import warpstr_amd.loci as wl
    try:
        fast5._libs()
    except fast5.Fast5Error as e:
        pytest.skip(str(e))
    src = os.path.join(GOLDEN, 'real', 'batch_0.fast5')
    ids = fast5.Fast5File(src).read_ids()[:10]
    monkeypatch.setattr(wl, 'STREAM_FROM_LOCI', 100)
    monkeypatch.setattr(wl, 'SHARED_BATCH_READS', 64)
    a, b = (_fast5_loci(str(tmp_path / t), src, ids, n_loci=230) for t in 'ab')
    tm_b = {}
    main_wrapper_loci(b, 1, _engine=FakeEngine, quiet=True, timings=tm_b)
    assert 'streamed' not in str(tm_b.get('reader_mode'))
    added = []

    class Engine(VbzFakeEngine):
        def add_automata(self, tables, flank_lengths):
            added.append(len(tables))
            return super().add_automata(tables, flank_lengths)
    tm = {}
    tables = main_wrapper_loci(a, threads, _engine=Engine, quiet=True, timings=tm)
    assert tm['reader_mode'].endswith('streamed with the set-up') and tm['vbz_batches'] >= 4
    assert tm['n_loci'] == 230 and tm['n_reads'] == sum(1 + li % 3 for li in range(230))
    assert len(tables) == 230 and len(tables[229][0]) == 1 + 229 % 3
    for la, lb in zip(a, b):
        for rel in OUTPUTS:
            assert filecmp.cmp(os.path.join(la.path, rel), os.path.join(lb.path, rel), shallow=False), rel
    # (whether automata arrived in flight depends on the race between the set-up and the first batch; when they did, all arrived)
    assert sum(added) == tm.get('automata_added_in_flight', 0) and sum(added) < 2 * 230


def test_a_streamed_run_raises_what_its_set_up_raises(tmp_path, monkeypatch):
    """A locus without its overview.csv in the middle of a streamed run: upstream's error for it, no batch left in flight, no
    reader thread left behind."""
    import threading
    from tests.helpers import GOLDEN
    from warpstr_amd import fast5
    import warpstr_amd.loci as wl
    try:
        fast5._libs()
    except fast5.Fast5Error as e:
        pytest.skip(str(e))
    src = os.path.join(GOLDEN, 'real', 'batch_0.fast5')
    ids = fast5.Fast5File(src).read_ids()[:10]
    monkeypatch.setattr(wl, 'STREAM_FROM_LOCI', 100)
    loci = _fast5_loci(str(tmp_path / 'a'), src, ids, n_loci=200)
    os.unlink(os.path.join(loci[150].path, 'overview.csv'))
    before = threading.active_count()
    with pytest.raises(FileNotFoundError, match='Not found the overview file'):
        main_wrapper_loci(loci, 1, _engine=VbzFakeEngine, quiet=True)
    assert threading.active_count() <= before


@pytest.mark.parametrize('threads', [1, 3])
def test_a_streamed_run_raises_what_collecting_a_batch_raises(tmp_path, threads, monkeypatch):
    """The batches' results are collected on a thread of their own: what that thread raises ends the run -- the submitting thread
    raises it, the reader thread and the reader processes stop, nothing is left behind -- whether the error comes with the first
    batch or in the middle of the run."""
    import threading
    from tests.helpers import GOLDEN
    from warpstr_amd import fast5
    import warpstr_amd.loci as wl
    try:
        fast5._libs()
    except fast5.Fast5Error as e:
        pytest.skip(str(e))
    src = os.path.join(GOLDEN, 'real', 'batch_0.fast5')
    ids = fast5.Fast5File(src).read_ids()[:10]
    monkeypatch.setattr(wl, 'STREAM_FROM_LOCI', 100)
    monkeypatch.setattr(wl, 'SHARED_BATCH_READS', 64)
    for fail_at in (0, 3):
        loci = _fast5_loci(str(tmp_path / f'a{fail_at}'), src, ids, n_loci=230)
        seen = []

        class Engine(VbzFakeEngine):
            def collect(self, ticket):
                seen.append(1)
                if len(seen) == fail_at + 1:
                    raise RuntimeError('the device decoders flagged a chunk')
                return super().collect(ticket)
        before = threading.active_count()
        with pytest.raises(RuntimeError, match='flagged a chunk'):
            main_wrapper_loci(loci, threads, _engine=Engine, quiet=True)
        assert threading.active_count() <= before and len(seen) == fail_at + 1


class ZstdFakeEngine(VbzFakeEngine):
    """... and one that also undoes the chunks' zstd frames itself (HipEngine since round 6: wsx_zstd_decode)."""
    DEVICE_ZSTD = True


@pytest.mark.parametrize('threads', [1, 3])
def test_readers_leave_zstd_to_an_engine_that_decodes_it(tmp_path, threads, monkeypatch):
    """An engine with DEVICE_ZSTD: the readers hand over every chunk whose frame the device decoder takes as the frame lies in
    the file (kinds 3 / 4 of the block table, the declared content size beside it) -- all of the upstream file's -- and the run's
    files are the same as with the frames undone by the readers, in one process and on reader processes, natively and through
    the ctypes reader."""
    from tests.helpers import GOLDEN
    from warpstr_amd import fast5, _readers
    try:
        fast5._libs()
    except fast5.Fast5Error as e:
        pytest.skip(str(e))
    src = os.path.join(GOLDEN, 'real', 'batch_0.fast5')
    ids = fast5.Fast5File(src).read_ids()[:10]
    a, b, c = (_fast5_loci(str(tmp_path / t), src, ids, n_loci=70) for t in 'abc')
    n_reads = sum(1 + li % 3 for li in range(70))
    tm_a, tm_b, tm_c = {}, {}, {}
    main_wrapper_loci(b, threads, _engine=VbzFakeEngine, quiet=True, timings=tm_b)
    main_wrapper_loci(a, threads, _engine=ZstdFakeEngine, quiet=True, timings=tm_a)
    assert tm_b['zstd_frames'] == 0 and tm_a['zstd_frames'] == n_reads and 'zstd and VBZ decoded on the GPU' in tm_a['reader_mode']
    monkeypatch.setenv('WARPSTR_NO_NATIVE_READER', '1')   # (reader processes started from here on inherit it)
    _readers._NATIVE = False
    try:
        main_wrapper_loci(c, threads, _engine=ZstdFakeEngine, quiet=True, timings=tm_c)
    finally:
        _readers._NATIVE = False
    assert tm_c['zstd_frames'] == n_reads
    for la, lb, lc in zip(a, b, c):
        for rel in OUTPUTS:
            assert filecmp.cmp(os.path.join(la.path, rel), os.path.join(lb.path, rel), shallow=False), rel
            assert filecmp.cmp(os.path.join(lc.path, rel), os.path.join(lb.path, rel), shallow=False), rel


@pytest.mark.parametrize('native', [True, False])
def test_a_locus_rows_as_one_item_expand_to_the_per_read_items(tmp_path, native):
    """What the streamed run sends its readers -- a locus's rows in one piece (LocusJob.rows_for_readers) -- becomes, in the
    reader (_readers.expand), exactly the items the run used to put together itself: upstream's path of a read's annotated file
    (src/caller/wrapper.py:49-50), the multi-read fall-back, the read's name; with and without a fast5_path / run_id column,
    whole and in slices, beside items that are per-read already."""
    from warpstr_amd import _readers, loci as L
    from warpstr_amd.caller import CallerConfig
    from warpstr_amd.pore_model import default_pore_model
    src = str(tmp_path / 'reads.fast5')
    made = _fast5_loci(str(tmp_path), src, [f'read{k}' for k in range(10)], n_loci=6)
    plain, _ = _make_loci(str(tmp_path / 'plain'))     # run_id 0, no fast5_path column, unsaved rows at the end
    for locus in made + plain:
        job = L.LocusJob(locus, default_pore_model(), {}, CallerConfig(), native=native)
        want = [(job.fast5_of(k), str(job.fast5_path[k]) if job.fast5_path is not None else None, job.names[k]) for k in range(job.n)]
        assert _readers.expand([job.rows_for_readers(0, job.n)]) == want
        if job.n >= 2:
            got = _readers.expand([job.rows_for_readers(0, 1), ('/x/y.fast5', None, 'z'), job.rows_for_readers(1, job.n)])
            assert got == want[:1] + [('/x/y.fast5', None, 'z')] + want[1:]
    assert _readers.expand([('/a', '/b', 'c')]) == [('/a', '/b', 'c')]


def test_default_readers_leave_this_process_a_share_when_the_engine_decodes_zstd(monkeypatch):
    """default_readers: as many readers as host threads, capped by the CPUs the cgroup grants; an engine that decodes zstd itself gets
    the share less an eighth (two of sixteen: the submitting and the collecting thread want CPUs of their own)."""
    import warpstr_amd.loci as wl
    monkeypatch.setattr(wl, 'cpu_share', lambda: 16)
    assert [wl.default_readers(t) for t in (1, 4, 16, 64)] == [1, 4, 16, 16]
    assert [wl.default_readers(t, True) for t in (1, 2, 4, 8, 16, 64)] == [1, 2, 3, 7, 14, 14]
    monkeypatch.setattr(wl, 'cpu_share', lambda: 3)
    assert wl.default_readers(16, True) == 3 and wl.default_readers(16) == 3


def test_a_run_keeps_its_timeline_when_asked(tmp_path, monkeypatch):
    """timings={'timeline': []}: the streamed run's events -- set-up, batches handed to the readers and answered, submitted and
    collected, the handle closed, the outputs written -- with the seconds since the call, in order; without the key nothing is kept,
    and the helpers' keys are gone from the dict either way."""
    from tests.helpers import GOLDEN
    from warpstr_amd import fast5
    import warpstr_amd.loci as wl
    try:
        fast5._libs()
    except fast5.Fast5Error as e:
        pytest.skip(str(e))
    src = os.path.join(GOLDEN, 'real', 'batch_0.fast5')
    ids = fast5.Fast5File(src).read_ids()[:10]
    monkeypatch.setattr(wl, 'STREAM_FROM_LOCI', 100)
    monkeypatch.setattr(wl, 'SHARED_BATCH_READS', 64)
    loci = _fast5_loci(str(tmp_path / 'a'), src, ids, n_loci=150)
    tm = {'timeline': []}
    main_wrapper_loci(loci, 1, _engine=VbzFakeEngine, quiet=True, timings=tm)
    names = [name.split(' [cpu')[0] for name, _ in tm['timeline']]
    times = [t for _, t in tm['timeline']]
    assert '_t0' not in tm and '_c0' not in tm
    for must in ('the streamed run begins', 'part set up', 'set-up done', 'last batch collected', 'handle closed', 'all reads called', 'outputs written'):
        assert must in names, must
    assert any(n.startswith('batch 0 answered') for n in names)
    assert any(n.startswith('batch 0 handed to the readers') for n in names) and any(n.endswith('submitted') for n in names)
    assert names.index('last batch collected') < names.index('handle closed') < names.index('all reads called') < names.index('outputs written')
    assert all(t >= 0 for t in times) and times[-1] == max(times) and times[-1] <= tm['total_s'] + 1e-3   # (times are rounded to 0.1 ms)
    assert sum(n.endswith('collected') and n.startswith('reads') for n in names) == sum(n.endswith('submitted') for n in names)
    tm2 = {}
    main_wrapper_loci(_fast5_loci(str(tmp_path / 'b'), src, ids, n_loci=150), 1, _engine=VbzFakeEngine, quiet=True, timings=tm2)
    assert 'timeline' not in tm2 and '_t0' not in tm2


def _rank_fast5(rank, world, port, root, src, ids, out_dir, partition):
    import json

    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    loci = _fast5_loci(root, src, ids, n_loci=70) if rank == 0 else None
    dist.barrier()
    if rank != 0:
        pats = [('(AGC)', 16), ('(AAAT)', 30), ('(CAG)CAACAG(CCG)', 20), ('(GGCCCC)', 24)]
        loci = [LocusPath(os.path.join(root, f'locus{li}'), *pats[li % 4]) for li in range(70)]
    tm, msg = {}, 'ok'
    try:
        main_wrapper_loci(loci, 3, _engine=VbzFakeEngine, quiet=True, shard=True, partition=partition, timings=tm)
    except Exception as e:  # noqa: BLE001
        msg = f'{type(e).__name__}: {e}'
    with open(os.path.join(out_dir, f'rank{rank}.json'), 'w') as f:
        json.dump({'msg': msg, 'partition': tm.get('partition'), 'reader_mode': tm.get('reader_mode'), 'vbz_batches': tm.get('vbz_batches')}, f)
    dist.destroy_process_group()


@pytest.mark.parametrize('partition', ['reads', 'loci'])
def test_two_ranks_read_their_share_of_the_files_through_reader_arenas(tmp_path, partition):
    """fast5 files, two ranks, reader processes with arenas on each: by read every rank takes a share of every locus's reads -- rows
    that are not next to each other in their loci go to the readers as the pieces they form (rows_of) --, by locus whole loci; the
    files equal one rank's either way."""
    import json
    from tests.helpers import GOLDEN
    from warpstr_amd import fast5
    try:
        fast5._libs()
    except fast5.Fast5Error as e:
        pytest.skip(str(e))
    src = os.path.join(GOLDEN, 'real', 'batch_0.fast5')
    ids = fast5.Fast5File(src).read_ids()[:10]
    one = _fast5_loci(str(tmp_path / 'one'), src, ids, n_loci=70)
    main_wrapper_loci(one, 1, _engine=FakeEngine, quiet=True)
    world, root = 2, str(tmp_path / 'two')
    mp.spawn(_rank_fast5, args=(world, _free_port(), root, src, ids, str(tmp_path), partition), nprocs=world, join=True)
    got = [json.load(open(tmp_path / f'rank{r}.json')) for r in range(world)]
    assert all(g['msg'] == 'ok' and g['partition'] == partition for g in got), got
    assert all('arenas' in str(g['reader_mode']) and g['vbz_batches'] >= 1 for g in got), got
    for li, l1 in enumerate(one):
        for rel in OUTPUTS:
            assert filecmp.cmp(os.path.join(l1.path, rel), os.path.join(root, f'locus{li}', rel), shallow=False), (li, rel)
