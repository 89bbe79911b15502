"""Per-kernel breakdown on other shapes, device-resident input, one stream: run under rocprofv3 --kernel-trace.
Usage: exp_kernels.py <cfg1|cfg5|ngc|cfg2> [n_reads]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from warpstr_amd import synth
from warpstr_amd.caller import HipCaller, pack_signals
SHAPES = {'cfg2': ('(AGC)', 16, 1500), 'cfg5': ('((CAGG){CAGM})(CAGA)(CA)', 40, (500, 5000)), 'cfg1': ('(AAAT)', 110, (2271, 3701)),
          'ngc': ('(NGC)', 24, 1800)}
name = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
pat, fl, T = SHAPES[name]
locus = synth.make_locus(pat, fl, 1)
rng = np.random.default_rng(1)
base = []
for _ in range(128):
    rev = bool(rng.random() < 0.5)
    t = int(T) if np.isscalar(T) else int(rng.integers(T[0], T[1] + 1))
    hi = max(1, min(30, (t // 4 - 2 * fl - 12) // 14))
    base.append((synth.squiggle(locus, rev, t, rng, lo=1, hi=hi, sigma=0.0)[0], rev))
pick = rng.integers(0, len(base), size=n)
sigs = [base[i][0] for i in pick]
sig, off = pack_signals(sigs)
sig = sig + 0.25 * rng.standard_normal(len(sig))
aut = np.array([int(base[i][1]) for i in pick], dtype=np.int32)
dev = torch.device('cuda', 0)
dsig = torch.from_numpy(sig).to(dev)
res = torch.zeros((n, 56), dtype=torch.uint8, device=dev)
hip = HipCaller([locus.template, locus.reverse], [fl, fl], stream=torch.cuda.current_stream().cuda_stream)
hip.set_streams(int(os.environ.get('EXP_STREAMS', '1')))
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hip.call_device(dsig.data_ptr(), off, aut, res.data_ptr())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f'{name} S={locus.template.n_states}/{locus.reverse.n_states} {hip.kernel_name(0)} n={n} samples={len(sig)}: {dt*1e3:.2f} ms per call, {n/dt:.3g} reads/s', flush=True)
