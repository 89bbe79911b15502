"""wsx_vbz_decode alone at the size of a from_fast5 batch: 2 048 blocks of the upstream test file's ten reads (59-170 k samples each,
their real StreamVByte blocks), HIP-event time per launch against the bytes it has to move (block bytes in, 2 B per sample out).
Usage: prof_vbz.py [blocks] [launches]   (under rocprofv3 --kernel-trace --stats for the kernel's own duration)"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_vbz_oracle import real_blocks
from warpstr_amd import _lib, synth
from warpstr_amd.caller import HipCaller

n_blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 20
real = real_blocks()
locus = synth.make_locus('(AGC)', 16, 1)
stream = torch.cuda.Stream()
hip = HipCaller([locus.template, locus.reverse], [16, 16], stream=stream.cuda_stream)
blobs = [real[i % len(real)] for i in range(n_blocks)]
src = np.concatenate([np.concatenate([b[1], np.zeros(-len(b[1]) % 16, np.uint8)]) for b in blobs])
blocks = np.zeros(n_blocks, _lib.VBZ_BLOCK_DTYPE)
at = out = 0
for i, (_, blk, n, zz) in enumerate(blobs):
    blocks[i] = (at, len(blk), out, n, _lib.VBZ_SVB_ZIGZAG if zz else _lib.VBZ_SVB, n, 0)
    at += len(blk) + (-len(blk) % 16)
    out += n
with torch.cuda.stream(stream):
    src_d = torch.from_numpy(src).cuda()
    dst_d = torch.empty(out, dtype=torch.int16, device='cuda')
    st_d = torch.empty(n_blocks, dtype=torch.int32, device='cuda')
    for _ in range(3):
        hip.vbz_decode_device(src_d.data_ptr(), len(src), blocks, dst_d.data_ptr(), out, st_d.data_ptr())
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(launches + 1)]
    ev[0].record()
    for k in range(launches):
        hip.vbz_decode_device(src_d.data_ptr(), len(src), blocks, dst_d.data_ptr(), out, st_d.data_ptr())
        ev[k + 1].record()
    stream.synchronize()
ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(launches)]
assert int(st_d.sum()) == 0
# the device's samples against the host decoder, every block of the first ten
import oracle.vbz as ovbz
got = dst_d.cpu().numpy()
for i in range(10):
    b = blocks[i]
    assert np.array_equal(got[b['dst_offset']:b['dst_offset'] + b['n_samples']], ovbz.decode_block(blobs[i][1], blobs[i][2], blobs[i][3]))
algo = int(blocks['src_bytes'].sum()) + 2 * out
print(json.dumps({'blocks': n_blocks, 'samples': out, 'block_bytes': int(blocks['src_bytes'].sum()), 'algorithmic_bytes_per_launch': algo,
                  'ms_per_launch_mean': float(np.mean(ms)), 'ms_per_launch_min': float(np.min(ms)),
                  'GB_per_s': algo / np.mean(ms) / 1e6, 'frac_of_8TBps': algo / np.mean(ms) / 1e6 / 8000.0,
                  'samples_per_s': out / np.mean(ms) * 1e3}))
