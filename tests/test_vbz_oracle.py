"""oracle/vbz.py (the CPU restatement of the VBZ decoder the GPU tests check wsx_vbz_decode against) pinned: against the upstream
test file's samples as the product's host decoders read them -- which tests/test_fast5.py pins to the normalised segments recorded
from upstream --, against those host decoders on random streams, and round trips through its own encoder.  CPU only."""
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
