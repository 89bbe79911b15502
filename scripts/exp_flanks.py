"""Throughput of flank localisation (wsx_locate_flanks) on the shape upstream runs: 110-base flanks in ~12 k-base windows,
device-resident inputs.  Prints pairs/s and cell updates/s; run under rocprofv3 for per-kernel numbers.
Usage: exp_flanks.py [n_pairs] [text_len] [flank_len]"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from warpstr_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 12000
P = int(sys.argv[3]) if len(sys.argv) > 3 else 110
rng = np.random.default_rng(1)
text = rng.integers(0, 4, size=(n, N)).astype(np.uint8)
text = np.frombuffer(b'ACGT', dtype=np.uint8)[text]
starts = rng.integers(0, N - P, size=n)
pat = np.stack([text[r, s:s + P] for r, s in enumerate(starts)])
mut = rng.random(pat.shape) < 0.08
pat = np.where(mut, np.frombuffer(b'ACGT', dtype=np.uint8)[rng.integers(0, 4, size=pat.shape)], pat).astype(np.uint8)
toff = np.arange(n + 1, dtype=np.int64) * N
poff = np.arange(n + 1, dtype=np.int64) * P
dev = torch.device('cuda', 0)
dt_text, dt_pat = torch.from_numpy(text.reshape(-1)).to(dev), torch.from_numpy(pat.reshape(-1)).to(dev)
hits = torch.zeros((n, 56), dtype=torch.uint8, device=dev)
sc = _lib.WsxAlignScores(2, -3, -3, -3)
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _lib.check(lib.wsx_locate_flanks(0, C.c_void_p(st), _lib.WSX_MEM_DEVICE, C.c_void_p(dt_text.data_ptr()), _lib.ptr(toff),
                                     C.c_void_p(dt_pat.data_ptr()), _lib.ptr(poff), n, C.byref(sc), C.c_void_p(hits.data_ptr()), None, 0),
               'wsx_locate_flanks')
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
h = hits.cpu().numpy().view(_lib.FLANK_HIT_DTYPE).reshape(-1)
ok = int(((h['status'] == 0) & (np.abs(h['start'] - starts) <= 3)).sum())
print(f'{n} pairs, text {N}, flank {P}: {dt*1e3:.2f} ms per call, {n/dt:.3g} pairs/s, {n*N*P/dt:.3g} cell updates/s; '
      f'{ok} of {n} flanks found at their planted position', flush=True)
