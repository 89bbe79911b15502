"""wsx_zstd_decode alone at the size of a from_fast5 batch: n frames cycling through the upstream test file's ten chunks (their real
zstd frames), HIP-event time per launch; then the same with wsx_vbz_decode behind it.  Usage: prof_zstd.py [n_frames] [launches]"""
import json
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from warpstr_amd import _lib, fast5, synth  # noqa: E402
from warpstr_amd.caller import HipCaller  # noqa: E402

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 10
h, zs = fast5._libs()
real = []
with fast5.Fast5File(os.path.join(ROOT, 'tests', 'golden', 'real', 'batch_0.fast5')) as f:
    for rid in f.read_ids():
        d, n, prm, chunk_len = f._open_signal(rid)
        for _, _, buf, size, plain in f._chunks(d, n, chunk_len):
            frame = bytes(buf[4:size])
            real.append((np.frombuffer(frame, np.uint8), int(zs.ZSTD_getFrameContentSize(frame, len(frame))), n))
        h.H5Dclose(d)
locus = synth.make_locus('(AGC)', 16, 1)
dev = torch.device('cuda:0')
stream = torch.cuda.Stream(device=dev)
hip = HipCaller([locus.template, locus.reverse], [16, 16], stream=stream.cuda_stream)
blobs = [real[i % len(real)] for i in range(n_frames)]
table = np.zeros(n_frames, _lib.ZSTD_FRAME_DTYPE)
at = out = 0
parts = []
for i, (fr, m, n) in enumerate(blobs):
    pad = -len(fr) % 16
    parts += [fr, np.zeros(pad, np.uint8)]
    table[i] = (at, len(fr), out, m)
    at += len(fr) + pad
    out += m + (-m % 16)
src = np.concatenate(parts)
with torch.cuda.stream(stream):
    src_d = torch.from_numpy(src).to(dev)
    dst_d = torch.empty(out, dtype=torch.uint8, device=dev)
    scr_d = torch.empty(out, dtype=torch.uint8, device=dev)
    st_d = torch.empty(n_frames, dtype=torch.int32, device=dev)
    args = (src_d.data_ptr(), len(src), table, dst_d.data_ptr(), out, scr_d.data_ptr(), st_d.data_ptr())
    for _ in range(2):
        hip.zstd_decode_device(*args)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(launches + 1)]
    ev[0].record()
    for k in range(launches):
        hip.zstd_decode_device(*args)
        ev[k + 1].record()
    stream.synchronize()
ms = float(np.mean([ev[k].elapsed_time(ev[k + 1]) for k in range(launches)]))
ok = int(st_d.sum()) == 0
got = dst_d.cpu().numpy()
for i in range(min(10, n_frames)):
    fr, m, n = blobs[i]
    ref = np.empty(m, np.uint8)
    assert zs.ZSTD_decompress(ref.ctypes.data, m, fr.tobytes(), len(fr)) == m
    o = int(table[i]['dst_offset'])
    ok = ok and np.array_equal(got[o:o + m], ref)
print(json.dumps({'kernel': 'zstd_decode_kernel', 'frames_per_launch': n_frames, 'compressed_bytes': int(table['src_bytes'].sum()), 'content_bytes': int(table['dst_bytes'].sum()),
                  'launch_ms': ms, 'content_GB_per_s': float(table['dst_bytes'].sum()) / ms / 1e6, 'frames_per_s': n_frames / ms * 1e3,
                  'equal_to_libzstd': bool(ok), 'libzstd_one_core_ms_per_frame': 0.06}, indent=1))
hip.close()
