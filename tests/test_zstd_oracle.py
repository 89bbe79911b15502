"""oracle/zstd_oracle.c (the CPU restatement of a Zstandard frame decoder the device decoder is held against) pinned against
libzstd -- a third-party implementation that IS installed here: the frames of the upstream test file, and frames libzstd makes at
several levels from inputs that make it use every block type, literals mode and sequence-table mode.  CPU only."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from oracle import zstd as ozstd
from tests.helpers import GOLDEN
from warpstr_amd import fast5

try:
    _, ZS = fast5._libs()
    ZS.ZSTD_compressBound.restype = C.c_size_t
    ZS.ZSTD_compressBound.argtypes = [C.c_size_t]
    HAVE = True
except fast5.Fast5Error:
    HAVE = False
pytestmark = pytest.mark.skipif(not HAVE, reason='no libzstd on this machine')


def compress(data: bytes, level: int, checksum=False, no_size=False) -> bytes:
    src = np.frombuffer(data, np.uint8) if len(data) else np.zeros(0, np.uint8)
    cap = ZS.ZSTD_compressBound(len(src))
    dst = np.empty(cap, np.uint8)
    if checksum or no_size:
        ZS.ZSTD_createCCtx.restype = C.c_void_p
        ZS.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
        ZS.ZSTD_compress2.restype = C.c_size_t
        ZS.ZSTD_compress2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        ZS.ZSTD_freeCCtx.argtypes = [C.c_void_p]
        cctx = ZS.ZSTD_createCCtx()
        ZS.ZSTD_CCtx_setParameter(cctx, 100, level)            # ZSTD_c_compressionLevel
        ZS.ZSTD_CCtx_setParameter(cctx, 201, int(checksum))    # ZSTD_c_checksumFlag
        ZS.ZSTD_CCtx_setParameter(cctx, 200, int(not no_size))  # ZSTD_c_contentSizeFlag
        n = ZS.ZSTD_compress2(cctx, dst.ctypes.data, cap, src.ctypes.data, len(src))
        ZS.ZSTD_freeCCtx(cctx)
    else:
        # (through a prototype of its own: tests/helpers.py declares the library object's ZSTD_compress with other argument types)
        fn = C.CFUNCTYPE(C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int)(('ZSTD_compress', ZS))
        n = fn(dst.ctypes.data, cap, src.ctypes.data, len(src), level)
    assert not ZS.ZSTD_isError(n)
    return dst[:n].tobytes()


def libzstd_decode(frame: bytes, n: int) -> bytes:
    out = np.empty(max(n, 1), np.uint8)
    got = ZS.ZSTD_decompress(out.ctypes.data, n, bytes(frame), len(frame))   # (argtypes of warpstr_amd._h5core: the source as bytes)
    assert got == n
    return out[:n].tobytes()


def inputs():
    rng = np.random.default_rng(17)
    text = (b'the quick brown fox jumps over the lazy dog. ' * 40 + bytes(rng.integers(97, 123, size=300).astype(np.uint8))) * 60
    sig = np.cumsum(rng.integers(-40, 41, size=150000)).astype(np.int16)
    from oracle import vbz
    svb = vbz.svb_encode(vbz.values_from_samples(sig, True)).tobytes()
    yield 'empty', b''
    yield 'one byte', b'x'
    yield 'short', b'hello, hello, hello'
    yield 'zeros', bytes(300000)                                   # RLE blocks / long matches
    yield 'random', bytes(rng.integers(0, 256, size=200000).astype(np.uint8))   # raw blocks
    yield 'skewed bytes', bytes(np.minimum(rng.geometric(0.08, size=400000), 255).astype(np.uint8))   # Huffman literals, no matches
    yield 'text', text                                             # matches: FSE sequence tables, repeat offsets
    yield 'streamvbyte block', svb                                 # what a VBZ chunk holds
    yield 'mixed', text[:70000] + bytes(100000) + svb[:150000] + text[:50000]
    yield 'few symbols', bytes(rng.integers(0, 3, size=250000).astype(np.uint8))   # a tiny alphabet: direct weights
    yield 'periodic', bytes(range(256)) * 900
    # two frequent bytes and forty that occur twice each -- their codes are the longest the format has (11 bits) -- standing in a row:
    # what a decoder's bit container has to survive (six 11-bit look-ups between two refills)
    rare = np.arange(60, 100, dtype=np.uint8)
    body = rng.integers(0, 2, size=300000).astype(np.uint8)
    for at in (1000, 150000):
        body[at:at + 40] = rare if at == 1000 else rare[::-1]
    yield 'longest codes in a row', bytes(body)
    yield 'small skewed', bytes(np.minimum(rng.geometric(0.3, size=180), 255).astype(np.uint8))   # Huffman literals in ONE stream
    # a second block whose literals are one byte over and over (RLE literals): copies of pieces of the first, random, block with
    # a single 'a' between them
    first = bytes(rng.integers(0, 256, size=1 << 17).astype(np.uint8))
    cuts = rng.integers(0, (1 << 17) - 40, size=3000)
    yield 'rle literals', first + b''.join(first[c:c + 24] + b'a' for c in cuts)


@pytest.mark.parametrize('level', [1, 3, 9, 19, -5])
def test_oracle_equals_libzstd_on_frames_libzstd_makes(level):
    kinds = set()
    for name, data in inputs():
        frame = compress(data, level)
        got, blocks = ozstd.decode(frame, want_blocks=True)
        assert got == data, (name, level)
        assert ozstd.content_size(frame) == len(data)
        assert libzstd_decode(frame, len(data)) == data
        kinds.add(blocks > 1)
    assert kinds == {False, True}                                   # single- and multi-block frames were seen


def test_oracle_takes_checksums_window_descriptors_and_frames_without_a_size():
    data = (b'abcdefghij' * 5000 + bytes(range(200))) * 4
    for kw in ({'checksum': True}, {'no_size': True}, {'checksum': True, 'no_size': True}):
        frame = compress(data, 3, **kw)
        if kw.get('no_size'):
            assert ozstd.content_size(frame) == -1
        assert ozstd.decode(frame, cap=len(data) + 10) == data


def test_oracle_refuses_what_is_not_a_frame_and_what_is_cut_short():
    frame = compress(b'some bytes to compress, some bytes to compress' * 50, 3)
    with pytest.raises(ValueError, match='not a zstd frame'):
        ozstd.decode(b'\x00' + frame[1:], cap=10000)
    for cut in (3, 6, len(frame) // 2, len(frame) - 1):
        with pytest.raises(ValueError):
            ozstd.decode(frame[:cut], cap=10000)
    with pytest.raises(ValueError, match='no room'):
        ozstd.decode(frame, cap=10)


def test_oracle_decodes_the_upstream_files_chunks_like_libzstd():
    """Every VBZ chunk of the upstream test file: the frame behind its 4-byte header, decoded by the oracle and by libzstd."""
    h, _ = fast5._libs()
    n_frames = 0
    with fast5.Fast5File(os.path.join(GOLDEN, 'real', 'batch_0.fast5')) as f:
        for rid in f.read_ids():
            d, n, prm, chunk_len = f._open_signal(rid)
            try:
                for _, _, buf, size, plain in f._chunks(d, n, chunk_len):
                    assert not plain and struct.unpack_from('<I', buf, 0)[0] == 2 * n
                    frame = bytes(buf[4:size])
                    m = ozstd.content_size(frame)
                    got, blocks = ozstd.decode(frame, want_blocks=True)
                    assert len(got) == m and got == libzstd_decode(frame, m), rid
                    assert blocks >= 1
                    n_frames += 1
            finally:
                h.H5Dclose(d)
    assert n_frames == 10


def frame_features(b: bytes):
    """What a frame uses, read from its headers alone: {('block', type), ('literals', type, streams), ('sequences', LL / OF / ML mode),
    ('huffman weights', 'direct' / 'fse')}."""
    out = set()
    fhd = b[4]
    flag, single, did = fhd >> 6, (fhd >> 5) & 1, fhd & 3
    pos = 5 + (0 if single else 1) + [0, 1, 2, 4][did] + ([1 if single else 0, 2, 4, 8][flag])
    while True:
        bh = int.from_bytes(b[pos:pos + 3], 'little')
        pos += 3
        last, btype, bsize = bh & 1, (bh >> 1) & 3, bh >> 3
        out.add(('block', btype))
        if btype == 2:
            lh = b[pos]
            ltype, sf = lh & 3, (lh >> 2) & 3
            if ltype < 2:
                hl = 1 if sf in (0, 2) else (2 if sf == 1 else 3)
                regen = lh >> 3 if hl == 1 else ((lh >> 4) + (b[pos + 1] << 4) + ((b[pos + 2] << 12) if hl == 3 else 0))
                comp, streams = (regen if ltype == 0 else 1), 1
            else:
                hl = 3 if sf < 2 else sf + 2
                v = int.from_bytes(b[pos:pos + hl], 'little')
                w = 10 if sf < 2 else (14 if sf == 2 else 18)
                comp, streams = (v >> (4 + w)) & ((1 << w) - 1), (1 if sf == 0 else 4)
                if ltype == 2:
                    out.add(('huffman weights', 'direct' if b[pos + hl] >= 128 else 'fse'))
            out.add(('literals', ltype, streams))
            sp = pos + hl + comp
            nseq = b[sp]
            at = sp + 1
            if nseq >= 128:
                at = sp + (2 if nseq < 255 else 3)
            if nseq:
                m = b[at]
                out.add(('sequences', m >> 6, (m >> 4) & 3, (m >> 2) & 3))
            else:
                out.add(('sequences', None))
        pos += 1 if btype == 1 else bsize
        if last:
            return out


def test_the_corpus_covers_the_format():
    """The frames the oracle is pinned on use every block type, every literals mode (raw, RLE, Huffman with one and four streams,
    treeless), both forms of Huffman weights and every sequence-table mode (predefined, RLE, FSE-compressed, repeat)."""
    seen = set()
    for level in (1, 3, 9, 19, -5):
        for _, data in inputs():
            if data:
                seen |= frame_features(compress(data, level))
    assert {('block', 0), ('block', 1), ('block', 2)} <= seen
    assert {t for k, t, *_ in [f for f in seen if f[0] == 'literals']} == {0, 1, 2, 3}, seen
    assert ('literals', 2, 1) in seen and ('literals', 2, 4) in seen
    assert {('huffman weights', 'direct'), ('huffman weights', 'fse')} <= seen
    modes = {m for f in seen if f[0] == 'sequences' and f[1] is not None for m in f[1:]}
    assert modes == {0, 1, 2, 3}, modes
    assert ('sequences', None) in seen


def test_the_readers_look_at_a_frames_headers_takes_treeless_literals_behind_a_tree():
    """`Fast5Core.frame_for_device` (what a reader asks before it leaves a frame to the GPU): every frame of the corpus is one the
    device takes -- the ones with treeless literals too, their tree comes from an earlier block --; a frame whose first
    Huffman-coded block is turned treeless by hand has no tree to take and is refused, and the oracle calls it corrupt."""
    from warpstr_amd._h5core import Fast5Core
    treeless = 0
    for level in (1, 3, 9):
        for name, data in inputs():
            frame = compress(data, level)
            assert Fast5Core.frame_for_device(frame) == len(data), (name, level)   # (at most four blocks each)
            treeless += any(ft[0] == 'literals' and ft[1] == 3 for ft in frame_features(frame))
    assert treeless >= 2
    data = dict(inputs())['skewed bytes']
    frame = bytearray(compress(data, 3))
    fhd = frame[4]
    pos = 5 + (0 if fhd & 0x20 else 1) + ((1 if fhd & 0x20 else 0) if fhd >> 6 == 0 else (2, 4, 8)[(fhd >> 6) - 1])
    bh = int.from_bytes(frame[pos:pos + 3], 'little')
    assert (bh >> 1) & 3 == 2 and frame[pos + 3] & 3 == 2
    frame[pos + 3] |= 1
    assert Fast5Core.frame_for_device(bytes(frame)) is None
    with pytest.raises(Exception):
        ozstd.decode(bytes(frame))
