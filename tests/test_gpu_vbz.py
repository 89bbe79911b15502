"""wsx_vbz_decode (csrc/wsx_vbz.hip: StreamVByte -> zig-zag -> running sum on the device) against oracle/vbz.py, through the C ABI:
the upstream test file's ten reads, random streams of every code length and size, plain blocks, unaligned offsets, blocks whose
keys ask for more bytes than they have, descriptors the library must refuse."""
import os

import numpy as np
import pytest

from oracle import vbz
from warpstr_amd import _lib, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
    from warpstr_amd.caller import HipCaller
    locus = synth.make_locus('(AGC)', 16, 1)
    h = HipCaller([locus.template, locus.reverse], [16, 16])
    yield h
    h.close()


def decode(hip, blobs, specs, pad=0, status=True):
    """blobs: list of uint8 arrays; specs: (kind, n) or (kind, n wanted, n coded) per blob.  Lays the blobs out back to back behind
    `pad` bytes (so that they start at odd addresses), runs the library, returns (samples per block, status per block)."""
    import torch
    src = np.concatenate([np.zeros(pad, np.uint8)] + [np.asarray(b, np.uint8) for b in blobs]) if blobs else np.zeros(pad, np.uint8)
    blocks = np.zeros(len(blobs), _lib.VBZ_BLOCK_DTYPE)
    at, out_at = pad, 3
    for i, (b, spec) in enumerate(zip(blobs, specs)):
        kind, n, coded = spec if len(spec) == 3 else (spec[0], spec[1], spec[1])
        blocks[i] = (at, len(b), out_at, n, kind, coded, 0)
        at += len(b)
        out_at += n + (i % 3)          # (gaps between the outputs: nothing may be written into them)
    dev = torch.device('cuda', hip.device)
    src_d = torch.from_numpy(src).to(dev)
    dst_d = torch.full((out_at + 5,), 12345, dtype=torch.int16, device=dev)
    st_d = torch.full((max(len(blobs), 1),), -7, dtype=torch.int32, device=dev)
    hip.vbz_decode_device(src_d.data_ptr(), len(src), blocks, dst_d.data_ptr(), dst_d.numel(), st_d.data_ptr() if status else 0)
    hip.synchronize()
    dst = dst_d.cpu().numpy()
    outs = [dst[int(b['dst_offset']):int(b['dst_offset']) + int(b['n_samples'])] for b in blocks]
    covered = np.zeros(len(dst), bool)
    for b in blocks:
        covered[int(b['dst_offset']):int(b['dst_offset']) + int(b['n_samples'])] = True
    assert (dst[~covered] == 12345).all()
    return outs, st_d.cpu().numpy()[:len(blobs)]


def test_upstream_test_file_decoded_on_the_device(hip):
    from tests.test_vbz_oracle import HAVE_HDF5, real_blocks
    from warpstr_amd import fast5
    if not HAVE_HDF5:
        pytest.skip('no libhdf5/libzstd on this machine')
    real = real_blocks()
    outs, st = decode(hip, [blk for _, blk, _, _ in real], [(_lib.VBZ_SVB_ZIGZAG, n) for _, _, n, _ in real], pad=1)
    assert (st == 0).all()
    with fast5.Fast5File(os.path.join(os.path.dirname(__file__), 'golden', 'real', 'batch_0.fast5')) as f:
        for (rid, blk, n, zz), got in zip(real, outs):
            assert np.array_equal(got, vbz.decode_block(blk, n, zz)) and np.array_equal(got, f.raw_signal(rid))


def test_random_streams_of_every_shape(hip):
    rng = np.random.default_rng(5)
    blobs, specs, want = [], [], []
    for n in (0, 1, 2, 3, 4, 5, 63, 64, 255, 256, 257, 1023, 1024, 1025, 4095, 4096, 4097, 150001):
        for kind in (_lib.VBZ_SVB_ZIGZAG, _lib.VBZ_SVB, _lib.VBZ_PLAIN):
            sig = rng.integers(-32768, 32768, size=n).astype(np.int16)
            if n > 2000:   # realistic: small differences, mostly one-byte values, with runs of two- and three-byte ones
                sig = np.cumsum(rng.integers(-40, 41, size=n)).astype(np.int16)
                sig[1000:1100] += rng.integers(-20000, 20000, size=100).astype(np.int16)
            if kind == _lib.VBZ_PLAIN:
                blobs.append(sig.view(np.uint8))
            else:
                blobs.append(vbz.svb_encode(vbz.values_from_samples(sig, kind == _lib.VBZ_SVB_ZIGZAG)))
            specs.append((kind, n))
            want.append(sig)
    # values of all four byte lengths in one block, decoded without the zig-zag (the running sum wraps)
    vals = rng.integers(0, 2 ** np.array([7, 8, 15, 16, 17, 24, 31, 32])[rng.integers(0, 8, size=3001)], dtype=np.uint64)
    blobs.append(vbz.svb_encode(vals))
    specs.append((_lib.VBZ_SVB, len(vals)))
    want.append(vbz.samples_from_values(vals.astype(np.uint32), False))
    blobs.append(vbz.svb_encode(vals))
    specs.append((_lib.VBZ_SVB_ZIGZAG, len(vals)))
    want.append(vbz.samples_from_values(vals.astype(np.uint32), True))
    # rounds whose 1 024 values take all four bytes each: the whole 4 KB window of a round and, at odd addresses, its 16-byte tail
    big = rng.integers(2 ** 24, 2 ** 32, size=5000, dtype=np.uint64)
    blobs.append(vbz.svb_encode(big))
    specs.append((_lib.VBZ_SVB, len(big)))
    want.append(vbz.samples_from_values(big.astype(np.uint32), False))
    for pad in (0, 1, 7, 15):
        outs, st = decode(hip, blobs, specs, pad=pad, status=pad != 7)
        if pad != 7:
            assert (st == 0).all()
        for i, (got, w) in enumerate(zip(outs, want)):
            assert np.array_equal(got, w), (pad, i, specs[i])


def test_blocks_that_code_more_values_than_are_wanted(hip):
    """A dataset's last chunk: HDF5 hands the filter a whole chunk, the block codes chunk-length values, the dataset holds fewer --
    the first n_samples leave, whatever the lengths of the values behind them."""
    rng = np.random.default_rng(21)
    blobs, specs, want = [], [], []
    for coded, n in ((4096, 1), (4096, 1000), (4096, 4095), (1025, 1024), (5000, 0), (3000, 2999), (20000, 7777)):
        sig = np.cumsum(rng.integers(-3000, 3001, size=coded)).astype(np.int16)
        for zz in (True, False):
            blobs.append(vbz.svb_encode(vbz.values_from_samples(sig, zz)))
            specs.append((_lib.VBZ_SVB_ZIGZAG if zz else _lib.VBZ_SVB, n, coded))
            want.append(sig[:n])
    for pad in (0, 5):
        outs, st = decode(hip, blobs, specs, pad=pad)
        assert (st == 0).all()
        for i, (got, w) in enumerate(zip(outs, want)):
            assert np.array_equal(got, w), (pad, i, specs[i])


def test_keys_that_ask_for_more_bytes_than_the_block_has(hip):
    """Flagged per block, nothing outside the block is read, the blocks beside it are decoded as ever."""
    rng = np.random.default_rng(8)
    good = np.cumsum(rng.integers(-30, 31, size=5000)).astype(np.int16)
    blk = vbz.svb_encode(vbz.values_from_samples(good, True))
    n = 5000
    bad = blk.copy()
    bad[600:(n + 3) // 4] = 0xFF     # every value from number 2400 on claims four bytes: the data run out near value 3500
    outs, st = decode(hip, [blk, bad, blk], [(_lib.VBZ_SVB_ZIGZAG, n)] * 3, pad=3)
    assert st.tolist() == [0, 1, 0]
    assert np.array_equal(outs[0], good) and np.array_equal(outs[2], good)
    assert np.array_equal(outs[1][:2400], good[:2400])
    lens = vbz.svb_block_lengths(bad[:(n + 3) // 4], n)
    fits = int(np.searchsorted(np.cumsum(lens), len(bad) - (n + 3) // 4, side='right'))   # values whose bytes are there
    ref = vbz.svb_decode(np.concatenate([bad, np.zeros(4 * n, np.uint8)]), n).astype(np.uint32)
    ref[fits:] = 0
    # (a lane decodes the four values of its key byte or none: the first lane that runs over gives zeros for all four)
    ref[4 * (fits // 4):] = 0
    assert np.array_equal(outs[1], vbz.samples_from_values(ref, True))


def test_descriptors_the_library_refuses(hip):
    import torch
    dev = torch.device('cuda', hip.device)
    src = torch.zeros(1000, dtype=torch.uint8, device=dev)
    dst = torch.zeros(1000, dtype=torch.int16, device=dev)

    def call(*block, src_bytes=1000, dst_samples=1000, coded=None, reserved=0):
        blocks = np.zeros(1, _lib.VBZ_BLOCK_DTYPE)
        blocks[0] = block + (block[3] if coded is None else coded, reserved)
        hip.vbz_decode_device(src.data_ptr(), src_bytes, blocks, dst.data_ptr(), dst_samples)
    call(0, 1000, 0, 500, _lib.VBZ_PLAIN)             # fine
    call(0, 125 + 500, 500, 500, _lib.VBZ_SVB_ZIGZAG)  # fine: key area + a byte per value
    hip.synchronize()
    for block, kw in [((0, 1001, 0, 10, 0), {}), ((990, 20, 0, 5, 0), {}), ((-1, 10, 0, 5, 0), {}), ((0, 10, 996, 5, 0), {}),
                      ((0, 10, -1, 5, 0), {}), ((0, 10, 0, -1, 0), {}), ((0, 10, 0, 5, 3), {}), ((0, 9, 0, 5, 0), {}),
                      ((0, 124 + 500, 0, 500, 1), {}), ((0, 100, 0, 10, 0), {'src_bytes': 50}), ((0, 100, 0, 10, 0), {'dst_samples': 9}),
                      ((0, 1000, 0, 500, 1), {'coded': 499}), ((0, 1000, 0, 10, 1), {'coded': 801}), ((0, 1000, 0, 10, 1), {'reserved': 1})]:
        with pytest.raises(RuntimeError, match='wsx_vbz_decode'):
            call(*block, **kw)
    hip.vbz_decode_device(0, 0, np.zeros(0, _lib.VBZ_BLOCK_DTYPE), 0, 0)   # no blocks: nothing to do


def test_a_launch_of_many_random_blocks(hip):
    """600 blocks of random kinds, sizes (0 .. 40 000 values), code lengths, wanted prefixes and addresses in ONE launch -- the shape
    of a batch of reads -- every sample against the oracle."""
    rng = np.random.default_rng(2024)
    blobs, specs, want = [], [], []
    for i in range(600):
        coded = int(rng.choice([0, 1, 2, 3, 4, 5, 1023, 1024, 1025, int(rng.integers(6, 40000))]))
        n = coded if rng.random() < 0.7 else int(rng.integers(0, coded + 1))
        style = rng.integers(0, 3)
        if style == 0:      # a squiggle: small differences
            sig = np.cumsum(rng.integers(-60, 61, size=coded)).astype(np.int16)
        elif style == 1:    # anything: differences of every size, wrap-around
            sig = rng.integers(-32768, 32768, size=coded).astype(np.int16)
        else:               # long flat stretches with jumps
            sig = np.repeat(rng.integers(-3000, 3000, size=coded // 50 + 1), 50)[:coded].astype(np.int16)
        kind = int(rng.choice([_lib.VBZ_SVB_ZIGZAG, _lib.VBZ_SVB_ZIGZAG, _lib.VBZ_SVB, _lib.VBZ_PLAIN]))
        if kind == _lib.VBZ_PLAIN:
            blobs.append(sig.view(np.uint8).copy())
            specs.append((kind, n, n))
        else:
            blobs.append(vbz.svb_encode(vbz.values_from_samples(sig, kind == _lib.VBZ_SVB_ZIGZAG)))
            specs.append((kind, n, coded))
        want.append(sig[:n])
    outs, st = decode(hip, blobs, specs, pad=9)
    assert (st == 0).all()
    for i, (got, w) in enumerate(zip(outs, want)):
        assert np.array_equal(got, w), (i, specs[i])
