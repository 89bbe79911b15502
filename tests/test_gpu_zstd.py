"""wsx_zstd_decode (csrc/wsx_zstd.hip: zstd frames decoded on the device) against libzstd itself -- the frames of the upstream test
file's chunks, and frames libzstd makes at several levels from inputs that use every block type, literals mode and sequence-table
mode -- byte for byte; what the decoder leaves to the host says so in its status, and nothing outside a frame's place is written."""
import os
import struct

import numpy as np
import pytest

from tests.helpers import GOLDEN
from tests.test_zstd_oracle import HAVE, compress, frame_features, inputs
from warpstr_amd import _lib, fast5, synth
from warpstr_amd.caller import HipCaller

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not HAVE, reason='no libzstd on this machine')]


def _handle():
    import torch
    locus = synth.make_locus('(AGC)', 16, 1)
    stream = torch.cuda.Stream(device=torch.device('cuda:0'))
    return HipCaller([locus.template, locus.reverse], [16, 16], stream=stream.cuda_stream), stream


def decode_on_device(frames_and_sizes):
    """[(frame bytes, content size)] -> ([content or None per frame], status array): one call, the frames 16-byte aligned in src,
    their contents 16-byte aligned in dst with a guard of 0xA5 bytes around every one."""
    import torch
    hip, stream = _handle()
    dev = torch.device('cuda:0')
    src_parts, table, at, out = [], np.zeros(len(frames_and_sizes), _lib.ZSTD_FRAME_DTYPE), 0, 32
    for i, (frame, n) in enumerate(frames_and_sizes):
        pad = -len(frame) % 16
        src_parts.append(np.frombuffer(frame, np.uint8))
        src_parts.append(np.zeros(pad, np.uint8))
        table[i] = (at, len(frame), out, n)
        at += len(frame) + pad
        out += n + (-n % 16) + 32
    src = np.concatenate(src_parts) if src_parts else np.zeros(0, np.uint8)
    with torch.cuda.stream(stream):
        src_d = torch.from_numpy(src).to(dev)
        dst_d = torch.full((out,), 0xA5, dtype=torch.uint8, device=dev)
        scr_d = torch.empty(out, dtype=torch.uint8, device=dev)
        st_d = torch.full((max(len(table), 1),), 77, dtype=torch.int32, device=dev)
        hip.zstd_decode_device(src_d.data_ptr(), len(src), table, dst_d.data_ptr(), out, scr_d.data_ptr(), st_d.data_ptr())
        stream.synchronize()
    got, status = dst_d.cpu().numpy(), st_d.cpu().numpy()[:len(table)]
    hip.close()
    res, covered = [], np.zeros(out, bool)
    for i, (frame, n) in enumerate(frames_and_sizes):
        o = int(table[i]['dst_offset'])
        covered[o:o + n] = True
        res.append(got[o:o + n].tobytes() if status[i] == 0 else None)
    assert (got[~covered] == 0xA5).all(), 'bytes outside the frames\' places were written'
    return res, status


def test_the_upstream_files_chunks_decode_like_libzstd():
    h, zs = fast5._libs()
    frames, want = [], []
    with fast5.Fast5File(os.path.join(GOLDEN, 'real', 'batch_0.fast5')) as f:
        for rid in f.read_ids():
            d, n, prm, chunk_len = f._open_signal(rid)
            try:
                for _, _, buf, size, plain in f._chunks(d, n, chunk_len):
                    assert not plain and struct.unpack_from('<I', buf, 0)[0] == 2 * n
                    frame = bytes(buf[4:size])
                    m = zs.ZSTD_getFrameContentSize(frame, len(frame))
                    out = np.empty(m, np.uint8)
                    assert zs.ZSTD_decompress(out.ctypes.data, m, frame, len(frame)) == m
                    frames.append((frame, int(m)))
                    want.append(out.tobytes())
            finally:
                h.H5Dclose(d)
    got, status = decode_on_device(frames * 3)    # (thirty frames in one launch)
    assert (status == 0).all(), status
    for k, g in enumerate(got):
        assert g == want[k % len(want)], f'frame {k}: content differs from libzstd\'s'


@pytest.mark.parametrize('level', [1, 3, 9, 19, -5])
def test_frames_libzstd_makes_decode_like_libzstd(level):
    """Every input of the oracle's corpus (tests/test_zstd_oracle.py) compressed at `level`: status 0 and libzstd's bytes --
    treeless literals (a block coded with the Huffman tree of an earlier block) included: across the levels the corpus has them."""
    named = [(name, data) for name, data in inputs() if data]
    frames = [(compress(data, level), len(data)) for _, data in named]
    got, status = decode_on_device(frames)
    for (name, data), (frame, _), g, st in zip(named, frames, got, status):
        assert st == 0, (name, level, st)
        assert g == data, (name, level)


def test_treeless_literals_take_the_tree_of_an_earlier_block():
    """The corpus holds frames with treeless literals (or this test would prove nothing); one of them decodes with its treeless block
    listed apart from the block that brought the tree.  And a frame whose FIRST Huffman-coded block is made treeless by hand -- no
    tree to take -- is corrupt, on the device as for the host's look at the headers."""
    from warpstr_amd._h5core import Fast5Core
    seen = 0
    for level in (1, 3, 9):
        for name, data in inputs():
            if not data:
                continue
            frame = compress(data, level)
            feats = frame_features(frame)
            if any(ft[0] == 'literals' and ft[1] == 3 for ft in feats):
                seen += 1
                assert Fast5Core.frame_for_device(frame) == len(data), name
    assert seen >= 2, 'the corpus no longer has a frame with treeless literals'
    data = dict(inputs())['skewed bytes']
    frame = bytearray(compress(data, 3))
    # the first compressed block's literals header: type bits 2 -> 3 where it is Huffman-coded
    pos = 5 + (0 if frame[4] & 0x20 else 1) + ((1 if frame[4] & 0x20 else 0) if frame[4] >> 6 == 0 else (2, 4, 8)[(frame[4] >> 6) - 1])
    bh = int.from_bytes(frame[pos:pos + 3], 'little')
    assert (bh >> 1) & 3 == 2 and frame[pos + 3] & 3 == 2, 'the first block of this frame used to be Huffman-coded'
    assert Fast5Core.frame_for_device(bytes(frame)) == len(data)
    frame[pos + 3] |= 1
    assert Fast5Core.frame_for_device(bytes(frame)) is None
    _, status = decode_on_device([(bytes(frame), len(data))])
    assert status[0] == 2


def test_a_corrupt_frame_says_so_and_the_others_are_untouched():
    good = compress(b'the quick brown fox jumps over the lazy dog. ' * 3000, 3)
    n = len(b'the quick brown fox jumps over the lazy dog. ' * 3000)
    bad = bytearray(good)
    bad[len(bad) // 2] ^= 0x5A                     # (a flipped byte in the middle of the entropy-coded part)
    short = good[:len(good) - 7]
    wrong_size = (good, n - 1)                    # the header declares one byte more than the caller made room for
    got, status = decode_on_device([(good, n), (bytes(bad), n), (short, n), wrong_size, (b'\x00' * 40, 10), (good, n)])
    assert status[0] == 0 and status[5] == 0 and got[0] == got[5] == b'the quick brown fox jumps over the lazy dog. ' * 3000
    assert status[2] == 2 and status[3] == 2 and status[4] == 2
    assert status[1] in (0, 2)                     # (a flipped literal bit may still be a valid stream: then the content differs)
    if status[1] == 0:
        assert got[1] != got[0]


def test_no_frames_and_frames_of_no_content():
    """A call without frames is a success that launches nothing; a frame whose content is empty (libzstd writes one for b'') decodes
    to nothing, beside frames of one byte and of a few, without touching a neighbour's place."""
    hip, stream = _handle()
    assert hip.zstd_decode_device(0, 0, np.zeros(0, _lib.ZSTD_FRAME_DTYPE), 0, 0, 0, 0) is None
    hip.close()
    items = [b'', b'x', b'', b'hello, hello, hello', b'']
    frames = [(compress(d, 3), len(d)) for d in items]
    got, status = decode_on_device(frames)
    assert (status == 0).all(), status
    assert got == items
