"""The .fast5 reader and the caller-only input path against the upstream test case's own data files
(tests/golden/real/: batch_0.fast5 + example.csv are copies of test/test_input/test_run1/fast5s/batch_0.fast5 and
test/test_caller_only/example.csv; real_aaat.npz holds what the upstream functions made of them).  CPU only."""
import os
import struct

import numpy as np
import pytest

from tests.helpers import GOLDEN, load_case
from warpstr_amd import fast5, overview as ov
from warpstr_amd.genotyper import call_alleles
from warpstr_amd.signal_prep import process_raw
from warpstr_amd.wrapper import get_raw_workload, prepare_caller_only

REAL = os.path.join(GOLDEN, 'real')
try:
    fast5._libs()
    HAVE_HDF5 = True
except fast5.Fast5Error:
    HAVE_HDF5 = False
needs_hdf5 = pytest.mark.skipif(not HAVE_HDF5, reason='no libhdf5/libzstd on this machine')


def svb_encode(vals):
    keys, body = bytearray((len(vals) + 3) // 4), bytearray()
    for i, v in enumerate(vals):
        nb = max(1, (int(v).bit_length() + 7) // 8)
        keys[i // 4] |= (nb - 1) << (2 * (i % 4))
        body += int(v).to_bytes(nb, 'little')
    return bytes(keys) + bytes(body)


@pytest.mark.parametrize('n', [0, 1, 3, 4, 5, 257])
def test_streamvbyte_and_vbz_chunk_round_trip(n):
    rng = np.random.default_rng(n)
    sig = rng.integers(-30000, 30000, size=n).astype(np.int16)
    if n > 4:
        sig[:4] = [0, 1, -1, 255]
    delta = np.diff(np.concatenate([[0], sig.astype(np.int64)]))  # |delta| reaches 60000: 1..3-byte codes
    zz = [(int(d) << 1) ^ (int(d) >> 63) for d in delta]
    chunk = struct.pack('<I', 2 * n) + svb_encode([z & 0xFFFFFFFF for z in zz])
    out = fast5.vbz_decode_chunk(chunk, 2, True, 0, 0)
    assert out.dtype == np.int16 and np.array_equal(out, sig)
    vals = rng.integers(0, 2 ** 32, size=n, dtype=np.uint64)
    assert np.array_equal(fast5.streamvbyte_decode(np.frombuffer(svb_encode(vals), np.uint8), n), vals.astype(np.uint32))
    if n:
        with pytest.raises(fast5.Fast5Error):
            fast5.streamvbyte_decode(np.frombuffer(svb_encode(vals)[:-1], np.uint8), n)
    with pytest.raises(fast5.Fast5Error):
        fast5.vbz_decode_chunk(chunk, 2, True, 1, 0)  # 1-bit-key variant: refused, not mis-decoded


def test_c_decoder_equals_the_numpy_decoder(monkeypatch):
    """csrc/seam_helper.c's StreamVByte / zig-zag / running-sum loop (what fast5.py uses when the helper is built) against
    the NumPy decoder on random blocks: every code length, wrap-around of the int16 running sum, truncated blocks."""
    if fast5._vbz_c() is None:
        pytest.skip('warpstr_amd/_seam_helper.so is not built')
    rng = np.random.default_rng(3)
    for n in (0, 1, 2, 5, 64, 1000, 4097):
        for zigzag in (True, False):
            vals = rng.integers(0, 2 ** np.array([7, 8, 15, 16, 17, 24, 31, 32])[rng.integers(0, 8, size=n)], dtype=np.uint64)
            chunk = struct.pack('<I', 2 * n) + svb_encode([int(v) for v in vals])
            got = fast5.vbz_decode_chunk(chunk, 2, zigzag, 0, 0)
            with monkeypatch.context() as mp:
                mp.setattr(fast5, '_VBZ_C', None)
                want = fast5.vbz_decode_chunk(chunk, 2, zigzag, 0, 0)
            assert got.dtype == np.int16 and np.array_equal(got, want), (n, zigzag)
            if n:
                with pytest.raises(fast5.Fast5Error):
                    fast5.vbz_decode_chunk(chunk[:-1], 2, zigzag, 0, 0)
                with pytest.raises(fast5.Fast5Error):
                    fast5.vbz_decode_chunk(chunk[:4 + (n + 3) // 4 - 1], 2, zigzag, 0, 0)


@needs_hdf5
def test_reader_on_upstream_multi_read_file():
    z = load_case('real_aaat')
    with fast5.Fast5File(os.path.join(REAL, 'batch_0.fast5')) as f:
        ids = f.read_ids()
        assert sorted(ids) == sorted(str(n) for n in z['names']) and len(ids) == 10
        for i, name in enumerate(z['names']):
            raw = f.raw_signal(str(name))
            lo, hi, n = (int(v) for v in z['raw_span'][i])
            assert raw.dtype == np.int16 and raw.size == n
            # spike removal + whole-read normalisation + slice: bit-identical to what the upstream functions produced
            assert np.array_equal(process_raw(raw, (lo, hi)), z[f'r{i}_signal'])
        with pytest.raises(fast5.Fast5Error):
            f.raw_signal('no-such-read')
    with pytest.raises(fast5.Fast5Error):
        fast5.read_raw_signal(os.path.join(REAL, 'example.csv'))


@needs_hdf5
def test_prepare_caller_only_and_raw_workload(tmp_path):
    # the csv names the fast5 relative to the upstream checkout
    d = tmp_path / 'test' / 'test_input' / 'test_run1' / 'fast5s'
    d.mkdir(parents=True)
    os.symlink(os.path.join(REAL, 'batch_0.fast5'), d / 'batch_0.fast5')
    loci = prepare_caller_only(os.path.join(REAL, 'example.csv'), str(tmp_path / 'out'), base_dir=str(tmp_path))
    assert list(loci) == ['Human_STR_1108232']
    _, df = ov.load_overview(loci['Human_STR_1108232'])
    assert list(df.columns) == ['fast5_path', 'locus', 'reverse', 'l_start_raw', 'r_end_raw', 'run_id', 'saved']
    assert set(df['run_id']) == {'run_0'} and set(df['saved']) == {1}
    names, revs, raws, pos = get_raw_workload(df, loci['Human_STR_1108232'])
    z = load_case('real_aaat')
    assert names == [str(n) for n in z['names']] and revs == [bool(r) for r in z['reverse']]
    assert [p[0] for p in pos] == [int(v) for v in z['raw_span'][:, 0]]
    assert [len(r) for r in raws] == [int(v) for v in z['raw_span'][:, 2]]
    bad = tmp_path / 'bad.csv'
    bad.write_text('fast5_path,locus,read_name\nx,y,z\n')
    with pytest.raises(ValueError):
        prepare_caller_only(str(bad), str(tmp_path / 'out2'))
    with pytest.raises(FileNotFoundError):
        prepare_caller_only(os.path.join(REAL, 'example.csv'), str(tmp_path / 'out3'), base_dir=str(tmp_path / 'nowhere'))


def test_upstream_known_answer_from_golden_lengths():
    """README.md section 2: 'Allele lengths as given by WarpSTR: (44, 40)'.  The per-read lengths recorded from the
    upstream caller on these reads genotype to exactly that."""
    z = load_case('real_aaat')
    lens = [len(str(z[f'r{i}_seq'][1])) for i in range(int(z['n_reads']))]
    assert sorted(set(lens)) == [40, 44]
    gt = call_alleles(lens, random_state=0)
    assert gt.heterozygous and sorted(gt.alleles, reverse=True) == [44, 40]


@needs_hdf5
def test_a_reader_process_decodes_into_its_arena_without_numpy(tmp_path):
    """The arena path of a reader process (loci._WorkerPool -> _hostworker -> _readers.decode_arena -> _h5core.decode_to): the
    samples in the arena equal Fast5File.raw_signal's, and the worker has not imported NumPy to get there (what makes sixteen of
    them start in a tenth of the time); with the native decoder switched off the same call falls back to fast5.py's decoders."""
    import mmap
    from warpstr_amd import _readers, loci
    path = os.path.join(REAL, 'batch_0.fast5')
    with fast5.Fast5File(path) as f:
        ids = f.read_ids()
        want = [f.raw_signal(i) for i in ids]
    items = [(str(tmp_path / 'absent.fast5'), path, i) for i in ids]

    def loaded_modules(prefix):   # (only its name travels to the worker)
        raise AssertionError
    for no_native in ('', '1'):
        old = os.environ.pop('WARPSTR_NO_HOST_NATIVE', None)
        if no_native:
            os.environ['WARPSTR_NO_HOST_NATIVE'] = '1'
        try:
            pool = loci._WorkerPool(1)
            pool.procs
        finally:
            os.environ.pop('WARPSTR_NO_HOST_NATIVE', None)
            if old is not None:
                os.environ['WARPSTR_NO_HOST_NATIVE'] = old
        try:
            arena, cap, base, lens, _ = pool.submit(_readers.decode_arena, (0, 1, items[:6])).result()
            arena2, cap2, base2, lens2, _ = pool.submit(_readers.decode_arena, (0, 1, items[6:])).result()
            assert base == 0 and base2 == sum(lens) and arena2 == arena and lens + lens2 == [len(w) for w in want]
            with open(arena, 'rb') as fh:
                mm = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
            got = np.frombuffer(mm, dtype=np.int16, count=sum(lens + lens2)).copy()
            mm.close()
            assert np.array_equal(got, np.concatenate(want))
            numpy_loaded = pool.submit(loaded_modules, 'numpy').result()
            assert bool(numpy_loaded) == bool(no_native) or fast5._vbz_native() is None, numpy_loaded
        finally:
            pool.shutdown()
        assert not os.path.exists(arena)   # the worker removes its arenas when it goes


def write_plain_single_read_fast5(path, sig, deflate=0, chunk=0):
    """A single-read file as steps 1-2 of the reference leave them (Raw/Reads/Read_7/Signal), written through libhdf5 itself:
    contiguous, or chunked with gzip."""
    import ctypes as C
    h, _ = fast5._libs()
    hid = C.c_int64
    for fn, res, args in [('H5Fcreate', hid, [C.c_char_p, C.c_uint, hid, hid]), ('H5Gcreate2', hid, [hid, C.c_char_p, hid, hid, hid]),
                          ('H5Screate_simple', hid, [C.c_int, C.POINTER(C.c_uint64), C.c_void_p]),
                          ('H5Dcreate2', hid, [hid, C.c_char_p, hid, hid, hid, hid, hid]), ('H5Pcreate', hid, [hid]),
                          ('H5Pset_chunk', C.c_int, [hid, C.c_int, C.POINTER(C.c_uint64)]), ('H5Pset_deflate', C.c_int, [hid, C.c_uint]),
                          ('H5Dwrite', C.c_int, [hid, hid, hid, hid, hid, C.c_void_p])]:
        f = getattr(h, fn)
        f.restype, f.argtypes = res, args
    fid = h.H5Fcreate(path.encode(), 2, 0, 0)
    assert fid >= 0
    for g in ('Raw', 'Raw/Reads', 'Raw/Reads/Read_7'):
        h.H5Gclose(h.H5Gcreate2(fid, g.encode(), 0, 0, 0))
    sp = h.H5Screate_simple(1, (C.c_uint64 * 1)(len(sig)), None)
    pl = h.H5Pcreate(hid.in_dll(h, 'H5P_CLS_DATASET_CREATE_ID_g').value)
    if chunk:
        assert h.H5Pset_chunk(pl, 1, (C.c_uint64 * 1)(chunk)) >= 0
        if deflate:
            assert h.H5Pset_deflate(pl, deflate) >= 0
    i16 = hid.in_dll(h, 'H5T_NATIVE_SHORT_g').value
    d = h.H5Dcreate2(fid, b'Raw/Reads/Read_7/Signal', i16, sp, 0, pl, 0)
    assert d >= 0
    sig = np.ascontiguousarray(sig, dtype=np.int16)
    assert h.H5Dwrite(d, i16, 0, 0, 0, sig.ctypes.data_as(C.c_void_p)) >= 0
    h.H5Dclose(d)
    h.H5Pclose(pl)
    h.H5Sclose(sp)
    h.H5Fclose(fid)


@needs_hdf5
def test_core_reader_equals_the_array_reader_on_every_storage(tmp_path):
    """_h5core.Fast5Core.decode_to (an address) against Fast5File.raw_signal: the upstream VBZ multi-read file, and contiguous,
    chunked and gzip single-read files (libhdf5's own pipeline)."""
    from warpstr_amd import _h5core
    rng = np.random.default_rng(5)
    sig = rng.integers(-2000, 2000, size=5000).astype(np.int16)
    cases = [(os.path.join(REAL, 'batch_0.fast5'), None)]
    for k, kw in enumerate(({}, {'chunk': 777}, {'chunk': 1024, 'deflate': 4})):
        p = str(tmp_path / f's{k}.fast5')
        write_plain_single_read_fast5(p, sig, **kw)
        cases.append((p, sig))
    for path, expect in cases:
        with fast5.Fast5File(path) as f, _h5core.Fast5Core(path) as c:
            for rid in (f.read_ids() or [None]):
                want = f.raw_signal(rid)
                buf = np.full(len(want) + 8, 12345, dtype=np.int16)
                seen = []
                try:
                    n = c.decode_to(rid, lambda n: (seen.append(n), buf.ctypes.data + 8)[1])
                except _h5core.NeedsNumpy:
                    assert fast5._vbz_native() is None
                    continue
                assert n == len(want) == seen[0] and np.array_equal(buf[4:4 + n], want)
                assert (buf[:4] == 12345).all() and (buf[4 + n:] == 12345).all()
                assert c.signal_length(rid) == n
                if expect is not None:
                    assert np.array_equal(want, expect)


@needs_hdf5
@pytest.mark.parametrize('zigzag,level', [(True, 1), (False, 3), (True, 0)])
def test_vbz_datasets_of_several_chunks(tmp_path, zigzag, level):
    """Reads whose dataset is several VBZ chunks (tests/helpers.write_vbz_fast5: coded by oracle/vbz.py's encoder, written without
    the plugin): the last chunk codes a whole chunk of which the dataset holds a part, one chunk is stored with the filter skipped.
    Every way of reading them gives the samples: the array reader, its NumPy decoders alone, the core reader to an address, and the
    blocks the device decoder takes (checked here by the oracle)."""
    import ctypes
    from oracle import vbz
    from tests.helpers import write_vbz_fast5
    from warpstr_amd import _h5core, _readers
    rng = np.random.default_rng(int(zigzag) + 2 * level)
    reads = {'r0': np.cumsum(rng.integers(-200, 201, size=10000)).astype(np.int16),       # 2.44 chunks
             'r1': rng.integers(-32768, 32768, size=4096).astype(np.int16),               # exactly one chunk, every code length
             'r2': np.cumsum(rng.integers(-5, 6, size=1)).astype(np.int16),               # a single sample
             'r3': np.cumsum(rng.integers(-90, 91, size=12289)).astype(np.int16)}         # three chunks and one sample
    path = write_vbz_fast5(str(tmp_path / 'multi.fast5'), reads, 4096, zigzag, level, skip_filter_on=(1,))
    with fast5.Fast5File(path) as f:
        assert sorted(f.read_ids()) == sorted(reads)
        for rid, sig in reads.items():
            assert np.array_equal(f.raw_signal(rid), sig), rid
            assert f.signal_length(rid) == len(sig)
    with _h5core.Fast5Core(path) as c:
        for rid, sig in reads.items():
            if fast5._vbz_native() is None:   # (a build without _host_loci.so: the core leaves VBZ to the array reader's decoders)
                with pytest.raises(_h5core.NeedsNumpy):
                    c.decode_to(rid, lambda n: 0)
                with pytest.raises(_h5core.NeedsNumpy):
                    c.blocks_to(rid, lambda n: 0)
                continue
            buf = np.full(len(sig) + 4, 777, np.int16)
            assert c.decode_to(rid, lambda n: buf.ctypes.data + 4) == len(sig)
            assert np.array_equal(buf[2:-2], sig) and (buf[:2] == 777).all() and (buf[-2:] == 777).all()
            room = np.zeros(4 * len(sig) + 65536, np.uint8)
            at, offs = [0], []

            def place(nbytes):
                at[0] = (at[0] + 15) & ~15
                offs.append(at[0])
                at[0] += nbytes
                assert at[0] <= len(room)
                return room.ctypes.data + offs[-1]
            n, blocks = c.blocks_to(rid, place)
            assert n == len(sig) and len(blocks) == -(-len(sig) // 4096) and sum(b[2] for b in blocks) == n
            pieces = []
            for (kind, nbytes, ns, nv, _content), off in zip(blocks, offs):
                blk = room[off:off + nbytes]
                assert nv >= ns and (kind == _h5core.PLAIN) == (len(pieces) == 1 and len(blocks) > 1)
                pieces.append(blk.view(np.int16)[:ns] if kind == _h5core.PLAIN else vbz.decode_block(blk, nv, kind == _h5core.SVB_ZIGZAG)[:ns])
            assert np.array_equal(np.concatenate(pieces), sig)
    # the NumPy decoders alone (a build without _host_loci.so takes them)
    import subprocess
    import sys
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from warpstr_amd import fast5; f = fast5.Fast5File(%r); "
            "assert fast5._vbz_native() is None; np.save(%r, np.concatenate([f.raw_signal(r) for r in %r]))"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), path, str(tmp_path / 'np.npy'), sorted(reads)))
    subprocess.run([sys.executable, '-c', code], check=True, env=dict(os.environ, WARPSTR_NO_HOST_NATIVE='1'))
    assert np.array_equal(np.load(str(tmp_path / 'np.npy')), np.concatenate([reads[r] for r in sorted(reads)]))
    # ... and what a reader process hands the parent of a GPU run (several blocks per read)
    items = [(str(tmp_path / 'absent'), path, rid) for rid in reads]
    arena, cap, base, used, lens, table, _ = _readers.pack_arena(('test_vbz_chunks', 1, items))
    try:
        t = np.frombuffer(table, np.int64).reshape(-1, 7)
        assert lens == [len(reads[r]) for r in reads]
        assert len(t) == (sum(-(-len(s) // 4096) for s in reads.values()) if fast5._vbz_native() is not None else len(reads))
        view = np.memmap(arena, dtype=np.uint8, mode='r')
        got = {r: [] for r in range(len(reads))}
        for r, kind, off, nbytes, ns, nv, _content in t:
            blk = np.array(view[off:off + nbytes])
            got[int(r)].append(blk.view(np.int16)[:ns] if kind == 0 else vbz.decode_block(blk, int(nv), kind == 1)[:ns])
        for r, rid in enumerate(reads):
            assert np.array_equal(np.concatenate(got[r]), reads[rid])
    finally:
        _readers._drop_arenas('test_vbz_chunks')
    assert ctypes.sizeof(ctypes.c_int16) == 2


@needs_hdf5
def test_native_reader_loop_equals_the_ctypes_reader(tmp_path):
    """csrc/host_reader.cpp (wsh_reader_pack: one library call per chunk of reads, the chunk's bytes by pread at the address libhdf5
    names, the zstd frame decompressed straight into the arena) against the ctypes reader it replaces: the same arena bytes, the same
    block table and lengths -- on the upstream multi-read file, on reads of several VBZ chunks with a filter-skipped chunk, through a
    small arena that has to grow -- and it declines (the ctypes reader takes over, or raises) what it does not do: a gzip
    single-read file, a file that is not there."""
    from tests.helpers import write_vbz_fast5
    from warpstr_amd import _readers
    if _readers._native_reader() is None:
        pytest.skip('warpstr_amd/_host_loci.so is not built with the reader loop (or HDF5 < 1.10.5)')
    rng = np.random.default_rng(8)
    reads = {'a': np.cumsum(rng.integers(-200, 201, size=10000)).astype(np.int16), 'b': rng.integers(-32768, 32768, size=4096).astype(np.int16),
             'c': np.cumsum(rng.integers(-90, 91, size=12289)).astype(np.int16)}
    multi = write_vbz_fast5(str(tmp_path / 'multi.fast5'), reads, 4096, True, 1, skip_filter_on=(1,))
    real = os.path.join(REAL, 'batch_0.fast5')
    ids = fast5.Fast5File(real).read_ids()
    items = [(str(tmp_path / 'absent.fast5'), real, r) for r in ids] + [(str(tmp_path / 'absent.fast5'), multi, r) for r in reads]

    def run(region, native):
        old = os.environ.pop('WARPSTR_NO_NATIVE_READER', None)
        if not native:
            os.environ['WARPSTR_NO_NATIVE_READER'] = '1'
        _readers._NATIVE = False
        try:
            out = [_readers.pack_arena((region, 1, items[:7])), _readers.pack_arena((region, 1, items[7:]))]   # (the second behind the first)
        finally:
            os.environ.pop('WARPSTR_NO_NATIVE_READER', None)
            if old is not None:
                os.environ['WARPSTR_NO_NATIVE_READER'] = old
            _readers._NATIVE = False
        arena = out[-1][0]
        used = out[-1][2] + out[-1][3]
        return out, bytes(np.memmap(arena, dtype=np.uint8, mode='r')[:used])
    try:
        (n1, n2), bytes_native = run('native_loop', True)
        (c1, c2), bytes_ctypes = run('ctypes_loop', False)
        for a, b in ((n1, c1), (n2, c2)):
            assert a[2:6] == b[2:6]                      # first byte, bytes used, lengths, block table
        assert n2[2] == n1[2] + n1[3] or n2[2] == ((n1[2] + n1[3] + 15) & ~15) or n2[2] >= n1[3]
        # (bytes between blocks are alignment padding: compare block by block)
        for t in (np.frombuffer(n1[5], np.int64).reshape(-1, 7), np.frombuffer(n2[5], np.int64).reshape(-1, 7)):
            for _, kind, off, nbytes, ns, nv, _content in t:
                assert bytes_native[off:off + nbytes] == bytes_ctypes[off:off + nbytes]
        assert n1[4] + n2[4] == [len(fast5.Fast5File(real).raw_signal(r)) for r in ids] + [len(v) for v in reads.values()]
        # what it declines: gzip single-read file -> the ctypes reader's plain block; a missing file -> that reader's error
        gz = str(tmp_path / 'gz.fast5')
        sig = rng.integers(-2000, 2000, size=3000).astype(np.int16)
        write_plain_single_read_fast5(gz, sig, chunk=1024, deflate=4)
        assert _readers._pack_native('declines', 0, [(gz, None, 'x')]) is None
        out = _readers.pack_arena(('declines', 2, [(gz, None, 'x')]))
        assert out[4] == [3000] and np.array_equal(np.memmap(out[0], dtype=np.int16, mode='r')[out[2] // 2:out[2] // 2 + 3000], sig)
        assert _readers._pack_native('declines', 0, [(str(tmp_path / 'nowhere.fast5'), None, 'x')]) is None
        with pytest.raises(fast5.Fast5Error):
            _readers.pack_arena(('declines', 3, [(str(tmp_path / 'nowhere.fast5'), None, 'x')]))
    finally:
        _readers._drop_arenas()
