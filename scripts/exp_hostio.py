"""PCIe-inclusive rate of the host-buffer boundary (wsx_call_batch with WSX_MEM_HOST), BASELINE configs[2] shape."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from warpstr_amd import synth
from warpstr_amd.caller import HipCaller

locus = synth.make_locus('(AGC)AACAGCCGCCAC(CGC)', 19, 2024, max_states=64)
n, T = 100000, 2000
rng = np.random.default_rng(0)
tpl = np.stack([synth.squiggle(locus, False, T, rng, sigma=0.0)[0] for _ in range(256)])
sig = (tpl[rng.integers(0, 256, size=n)] + 0.25 * rng.standard_normal((n, T))).reshape(-1)
off = np.arange(n + 1, dtype=np.int64) * T
aut = np.zeros(n, np.int32)
hip = HipCaller([locus.template, locus.reverse], [19, 19], workspace_limit=96 << 30)
for rep in range(3):
    t0 = time.perf_counter()
    res, _ = hip.call(sig, off, aut)
    dt = time.perf_counter() - t0
    tm = hip.last_timing()
    print(f'host-buffer call: {dt*1e3:.1f} ms wall ({n/dt:.3g} reads/s), device region {tm["total_ms"]:.1f} ms, ok={int((res["status"]==0).sum())}', flush=True)
