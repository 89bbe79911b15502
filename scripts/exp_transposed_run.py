"""Runs build/exp/libfill_t.so (scripts/exp_transposed_gen.py) on the headline workload: last DP row against wsx_warp_batch for
the first reads, then the launch time for all reads.  Usage: exp_transposed_run.py <headline|cfg1> [reads]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from warpstr_amd.caller import HipCaller
from exp_transposed_gen import automaton
name = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
T = 2000
dev = torch.device('cuda', 0)
locus, values, preds, end = automaton(name)
wl = bench.make_headline(n, T, 1000, dev)
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'build/exp/libfill_t.so'))
lib.run_fill_t.restype = ctypes.c_float
lib.run_fill_t.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
nw, rpw = lib.fill_t_words(), lib.fill_t_rpw()
S = len(values)
waves = (n + rpw - 1) // rpw
bp = torch.zeros(waves * T * nw, dtype=torch.int64, device=dev)
last = torch.full((n, S), float('nan'), dtype=torch.float64, device=dev)
ms = lib.run_fill_t(wl.signal.data_ptr(), n, T, bp.data_ptr(), last.data_ptr(), S, 5)
torch.cuda.synchronize()
print(f'{name}: transposed fill, {n} reads x {T} rows, S={S}: best launch {ms:.3f} ms; mask bytes per read-row {nw * 8 / rpw:.1f}')
nc = min(n, 1024)
hip = HipCaller([locus.template], [bench.HEADLINE[1] if name == 'headline' else bench.CFG1[1]])
sig = wl.signal[:nc * T].cpu().numpy()
off = np.arange(nc + 1, dtype=np.int64) * T
ref = hip.warp(sig, off, np.zeros(nc, np.int32), want_last_row=True)['last_row'][:, :S]
got = last[:nc].cpu().numpy()
same = (ref == got) | (np.isinf(ref) & np.isinf(got))
print(f'last row vs wsx_warp_batch on {nc} reads: {int((~same).sum())} differing cells of {same.size}')
