"""Throughput of the GPU signal loader (wsx_prepare_signals): int16 raw reads resident in HBM -> spike removal, whole-read
MAD normalisation, slice to the STR segment as f64.  Usage: exp_prep.py [n_reads] [raw_len] [segment_len]"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from warpstr_amd import synth, _lib
from warpstr_amd.caller import HipCaller
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
T = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(1)
raw = (torch.randn((n, L), generator=g, device=dev) * 60 + 480).round().clamp(0, 2000).to(torch.int16)
spk = torch.rand((n, L), generator=g, device=dev) < 0.002          # a few spikes for brute_remove
raw = torch.where(spk, torch.full_like(raw, 1900), raw).reshape(-1).contiguous()
roff = np.arange(n + 1, dtype=np.int64) * L
lo = np.full(n, L // 2, np.int64); hi = lo + T - 1
ooff = np.arange(n + 1, dtype=np.int64) * T
out = torch.zeros(n * T, dtype=torch.float64, device=dev)
ss = torch.zeros((n, 2), dtype=torch.float64, device=dev)
locus = synth.make_locus('(AGC)', 16, 1)
hip = HipCaller([locus.template, locus.reverse], [16, 16], stream=torch.cuda.current_stream().cuda_stream)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _lib.check(hip.lib.wsx_prepare_signals(hip.handle, _lib.WSX_MEM_DEVICE, C.c_void_p(raw.data_ptr()), _lib.ptr(roff), _lib.ptr(lo),
                                           _lib.ptr(hi), n, 1, C.c_void_p(out.data_ptr()), _lib.ptr(ooff), C.c_void_p(ss.data_ptr())),
               'wsx_prepare_signals')
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f'{n} reads x {L} raw samples -> {T}-sample segments: {dt*1e3:.2f} ms per call, {n/dt:.3g} reads/s, '
      f'{n*L*2/dt/1e9:.0f} GB/s of raw int16 in (every sample is read twice and written once)', flush=True)
print('shift/scale of read 0:', ss[0].cpu().numpy(), 'segment mean/std:', float(out[:T].mean()), float(out[:T].std()))
