"""oracle/vbz.py (the CPU restatement of the VBZ decoder the GPU tests check wsx_vbz_decode against).  What is independent of this
repository's own reader: a block typed in by hand from the published format, and the structural consistency of the upstream test
file (frame sizes, chunk headers, the sequencer's `duration` attribute, the DAC range).  What is not: the comparison with the
product's host decoders on the upstream file and on random streams -- oracle and host decoders are two restatements by the same
authors, and tests/golden/real_aaat.npz was recorded through the host decoder (h5py + the filter plugin exist nowhere here), so
that part of the pin is end to end only: the decoded reads call to the README's (44, 40).  CPU only."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from oracle import vbz
from tests.helpers import GOLDEN
from warpstr_amd import fast5

REAL = os.path.join(GOLDEN, 'real')
try:
    fast5._libs()
    HAVE_HDF5 = True
except fast5.Fast5Error:
    HAVE_HDF5 = False


def real_blocks():
    """(read id, StreamVByte block, samples, zig-zag) of every read of the upstream test file: its one chunk, zstd undone."""
    h, zs = fast5._libs()
    out = []
    with fast5.Fast5File(os.path.join(REAL, 'batch_0.fast5')) as f:
        for rid in f.read_ids():
            d, n, vbz_params, chunk_len = f._open_signal(rid)
            try:
                assert vbz_params[:2] == [0, 2] and chunk_len >= n
                for start, want, buf, size, plain in f._chunks(d, n, chunk_len):
                    assert not plain and struct.unpack_from('<I', buf, 0)[0] == 2 * n
                    body = bytes(buf[4:size])
                    m = zs.ZSTD_getFrameContentSize(body, len(body))
                    blk = C.create_string_buffer(m)
                    assert zs.ZSTD_decompress(blk, m, body, len(body)) == m
                    out.append((rid, np.frombuffer(blk.raw[:m], np.uint8), n, bool(vbz_params[2])))
            finally:
                f.h.H5Dclose(d)
    return out


@pytest.mark.skipif(not HAVE_HDF5, reason='no libhdf5/libzstd on this machine')
def test_oracle_reads_the_upstream_test_file_like_the_host_decoders():
    with fast5.Fast5File(os.path.join(REAL, 'batch_0.fast5')) as f:
        for rid, blk, n, zz in real_blocks():
            assert zz
            assert np.array_equal(vbz.decode_block(blk, n, zz), f.raw_signal(rid))


def test_oracle_round_trips_and_equals_the_numpy_decoder_on_random_streams():
    rng = np.random.default_rng(11)
    for n in (0, 1, 2, 3, 4, 5, 255, 256, 257, 1023, 1024, 1025, 5000):
        for zz in (True, False):
            sig = rng.integers(-32768, 32768, size=n).astype(np.int16)        # differences of every size, wrap-around included
            if n > 8:
                sig[:8] = [0, 1, -1, 127, -128, 255, 32767, -32768]
            blk = vbz.svb_encode(vbz.values_from_samples(sig, zz))
            assert np.array_equal(vbz.decode_block(blk, n, zz), sig), (n, zz)
            chunk = struct.pack('<I', 2 * n) + blk.tobytes()
            assert np.array_equal(fast5.vbz_decode_chunk(chunk, 2, zz, 0, 0), sig), (n, zz)
    vals = rng.integers(0, 2 ** 32, size=1000, dtype=np.uint64)
    assert np.array_equal(vbz.svb_decode(vbz.svb_encode(vals), 1000), vals.astype(np.uint32))
    assert np.array_equal(vbz.svb_decode(vbz.svb_encode(vals), 1000), fast5.streamvbyte_decode(vbz.svb_encode(vals), 1000))
    with pytest.raises(ValueError):
        vbz.svb_decode(vbz.svb_encode(vals)[:-1], 1000)
    with pytest.raises(ValueError):
        vbz.svb_decode(vbz.svb_encode(vals)[:200], 1000)


def test_hand_written_streamvbyte_vector_from_the_format_text():
    """A block typed in as bytes from the published layout (StreamVByte: one key byte per four values, two bits per value = byte
    length - 1, the first value in the LOW bits, little-endian value bytes back to back; VBZ version 0: the values are zig-zag
    mapped differences of consecutive samples, (d << 1) ^ (d >> 31), the first against 0) -- NOT produced by svb_encode or any
    other code of this repository.  All four byte lengths, sign changes, the largest step an int16 signal can take (32767 ->
    -32768: d = -65535, a 3-byte value) and a 4-byte value whose difference wraps in the sample type (d = 0x800002: +2 mod 2^16),
    and a last key byte that describes one value only.

        sample   3     2   130  -200   32767  -32768  -32768      -32766  -32765
        d        3    -1   128  -330   32967  -65535       0  0x00800002       1
        zig-zag  6     1   256   659   65934  131069       0  0x01000004       2
        bytes    1     1     2     2       3       3       1           4       1
        code     0     0     1     1       2       2       0           3       0
    keys: 0 | 0<<2 | 1<<4 | 1<<6 = 0x50;  2 | 2<<2 | 0<<4 | 3<<6 = 0xCA;  0 = 0x00"""
    block = bytes([0x50, 0xCA, 0x00,
                   0x06,
                   0x01,
                   0x00, 0x01,
                   0x93, 0x02,
                   0x8E, 0x01, 0x01,
                   0xFD, 0xFF, 0x01,
                   0x00,
                   0x04, 0x00, 0x00, 0x01,
                   0x02])
    want = np.array([3, 2, 130, -200, 32767, -32768, -32768, -32766, -32765], np.int16)
    arr = np.frombuffer(block, np.uint8)
    assert vbz.svb_decode(arr, 9).tolist() == [6, 1, 256, 659, 65934, 131069, 0, 0x01000004, 2]
    assert np.array_equal(vbz.decode_block(arr, 9, True), want)
    # ... and the product's host decoders on the same bytes (NumPy form, C loop, native library): an uncompressed VBZ chunk
    chunk = struct.pack('<I', 18) + block
    assert np.array_equal(fast5.vbz_decode_chunk(chunk, 2, True, 0, 0), want)
    assert fast5.streamvbyte_decode(arr, 9).tolist() == [6, 1, 256, 659, 65934, 131069, 0, 0x01000004, 2]
    # without the zig-zag flag the values are the differences themselves (three values, key 0 | 0<<2 | 1<<4): 6, then +1, +256
    assert vbz.decode_block(np.frombuffer(bytes([0x10, 0x06, 0x01, 0x00, 0x01]), np.uint8), 3, False).tolist() == [6, 7, 263]
    # the encoder the other tests build their streams with must produce exactly these bytes
    assert vbz.svb_encode(np.array([6, 1, 256, 659, 65934, 131069, 0, 0x01000004, 2], np.uint64)).tobytes() == block


def _raw_attr_u64(h, fid, group: str, name: str) -> int:
    """An integer attribute of a group through the HDF5 C API (ctypes), read as a native uint64."""
    hid = C.c_int64
    h.H5Aopen_by_name.restype, h.H5Aopen_by_name.argtypes = hid, [hid, C.c_char_p, C.c_char_p, hid, hid]
    h.H5Aread.restype, h.H5Aread.argtypes = C.c_int, [hid, hid, C.c_void_p]
    h.H5Aclose.restype, h.H5Aclose.argtypes = C.c_int, [hid]
    a = h.H5Aopen_by_name(fid, group.encode(), name.encode(), 0, 0)
    assert a >= 0, (group, name)
    try:
        v = C.c_uint64(0)
        assert h.H5Aread(a, hid.in_dll(h, 'H5T_NATIVE_UINT64_g').value, C.byref(v)) >= 0
        return int(v.value)
    finally:
        h.H5Aclose(a)


@pytest.mark.skipif(not HAVE_HDF5, reason='no libhdf5/libzstd on this machine')
def test_structure_of_the_upstream_file_is_consistent_with_the_decoder():
    """What the upstream test file says about its own reads WITHOUT any decoder of this repository (the pin of oracle/vbz.py is
    otherwise end to end only: see its docstring): for every chunk, the zstd frame's declared content size equals the key area
    plus the value bytes the keys announce, to the byte; the chunk's u32 header equals 2 x the dataset's length; the `duration`
    attribute the sequencer wrote into the read's Raw group equals the number of samples decoded; the decoded samples are DAC
    values of a 11-bit-range R9.4 device (0..2047 here -- a wrong byte order, key order or sign rule leaves that range at once)."""
    h, _ = fast5._libs()
    blocks = real_blocks()
    assert len(blocks) == 10
    with fast5.Fast5File(os.path.join(REAL, 'batch_0.fast5')) as f:
        for rid, blk, n, zz in blocks:
            n_keys = (n + 3) // 4
            lens = vbz.svb_block_lengths(blk[:n_keys], n)
            assert len(blk) == n_keys + int(lens.sum()), rid           # frame content size == keys + announced value bytes, exactly
            # (the key bits past the last value are zero, as the encoder leaves them)
            tail = blk[n_keys - 1] >> (2 * (n - 4 * (n_keys - 1))) if n % 4 else 0
            assert tail == 0, rid
            assert f.signal_length(rid) == n
            assert _raw_attr_u64(h, f.fid, f'read_{rid}/Raw', 'duration') == n, rid
            samples = vbz.decode_block(blk, n, zz)
            assert len(samples) == n and samples.min() >= 0 and samples.max() <= 2047, (rid, samples.min(), samples.max())
            # differences of a nanopore squiggle are small: nearly all values take one byte, none takes four
            assert (lens == 1).mean() > 0.7 and (lens == 4).sum() == 0, rid
