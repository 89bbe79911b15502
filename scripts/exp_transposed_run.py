"""Runs build/exp/libfill_t.so (scripts/exp_transposed_gen.py) on the headline workload: last DP row against wsx_warp_batch for
the first reads, then the launch time for all reads.  Usage: exp_transposed_run.py <headline|cfg1> [reads]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from warpstr_amd.caller import HipCaller
from exp_transposed_gen import automaton
name = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 100000
T = 2000
dev = torch.device('cuda', 0)
locus, values, preds, end = automaton(name)
wl = bench.make_headline(n, T, 1000, dev)
masked = '--masked' in sys.argv
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'build/exp/libfill_tm.so' if masked else 'build/exp/libfill_t.so'))
lib.run_fill_t.restype = ctypes.c_float
lib.run_fill_t.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
nw, rpw = lib.fill_t_words(), lib.fill_t_rpw()
S = len(values)
waves = (n + rpw - 1) // rpw
bp = torch.zeros(waves * T * nw, dtype=torch.int64, device=dev)
last = torch.full((n, S), float('nan'), dtype=torch.float64, device=dev)
mask = None
if masked:  # per-read masks in runs, ~58 % of the samples (what the bad-repeat mask covers on this workload)
    g = torch.Generator(device=dev); g.manual_seed(5)
    blocks = (torch.rand((n, T // 40 + 1), generator=g, device=dev) < 0.58)
    mask = blocks.repeat_interleave(40, dim=1)[:, :T].contiguous().to(torch.uint8)
    mask = torch.roll(mask, shifts=7, dims=1).contiguous()
    if '--mask0' in sys.argv: mask.zero_()
    if '--mask1' in sys.argv: mask.fill_(1)
    if '--mask-late' in sys.argv: mask[:, :1000] = 0
ms = lib.run_fill_t(wl.signal.data_ptr(), n, T, bp.data_ptr(), last.data_ptr(), S, 5, mask.data_ptr() if masked else None)
torch.cuda.synchronize()
print(f'{name}: transposed fill, {n} reads x {T} rows, S={S}: best launch {ms:.3f} ms; mask bytes per read-row {nw * 8 / rpw:.1f}')
nc = min(n, 1024)
hip = HipCaller([locus.template], [10])  # flank_length 10 for the reference: the corner cut (rows > T - 6 (fl - 10)) is then empty, as in the experiment
sig = wl.signal[:nc * T].cpu().numpy()
off = np.arange(nc + 1, dtype=np.int64) * T
ref = hip.warp(sig, off, np.zeros(nc, np.int32), mask=(mask[:nc].cpu().numpy().reshape(-1) if masked else None), want_last_row=True)['last_row'][:, :S]
got = last[:nc].cpu().numpy()
same = (ref == got) | (np.isinf(ref) & np.isinf(got))
fin = np.ones_like(same)
bad = ~same
if bad.any():
    r0 = int(np.argwhere(bad)[0][0]); print('first differing read', r0, 'states', np.flatnonzero(bad[r0])[:12], 'got', got[r0][bad[r0]][:3], 'ref', ref[r0][bad[r0]][:3])
print(f'last row vs wsx_warp_batch on {nc} reads ({"masked" if masked else "unmasked"} pass): {int((~same & fin).sum())} differing cells of {int(fin.sum())}')
