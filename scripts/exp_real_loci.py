"""The loci of upstream's example / test configurations at the default flank length (110), whole call (both passes),
device-resident input, pipelined calls: which fill kernel each gets and what a call of n reads x T samples takes.
Usage: exp_real_loci.py [n_reads] [samples]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from warpstr_amd import synth, _lib
from warpstr_amd.caller import HipCaller

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
dev = torch.device('cuda', 0)
LOCI = [('AAAT (test/test_caller_only)', '(AAAT)'), ('HD (example/config.yaml)', '(AGC)AACAGCCGCCAC(CGC)'),
        ('DM2 (example/config.yaml)', '((CAGG){CAGM})(CAGA)(CA)'), ('simple (AGC)', '(AGC)'), ('(GGCCCC)', '(GGCCCC)')]
print(f'{n} reads x {T} samples, flank 110, both passes, device-resident, 8 pipelined calls', flush=True)
for name, pat in LOCI:
    fl = 110
    loc = synth.make_locus(pat, fl, 7)
    rng = np.random.default_rng(5)
    base = []
    for _ in range(96):
        rev = bool(rng.random() < 0.5)
        hi = max(1, min(30, (T // 4 - 2 * fl - 12) // 14))
        base.append((synth.squiggle(loc, rev, T, rng, lo=1, hi=hi, sigma=0.0)[0], rev))
    pick = rng.integers(0, len(base), size=n)
    clean = torch.from_numpy(np.stack([b[0] for b in base])).to(dev)
    g = torch.Generator(device=dev); g.manual_seed(3)
    sig = (clean[torch.from_numpy(pick).to(dev)] + 0.25 * torch.randn((n, T), generator=g, device=dev, dtype=torch.float64)).reshape(-1).contiguous()
    aut = np.array([int(base[i][1]) for i in pick], np.int32)
    off = np.arange(n + 1, dtype=np.int64) * T
    res = [torch.zeros((n, 56), dtype=torch.uint8, device=dev) for _ in range(2)]
    hip = HipCaller([loc.template, loc.reverse], [fl, fl], stream=torch.cuda.current_stream().cuda_stream)
    hip.set_pipelined(True)
    for k in range(2):
        hip.call_device(sig.data_ptr(), off, aut, res[k & 1].data_ptr())
    hip.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(8):
        hip.call_device(sig.data_ptr(), off, aut, res[k & 1].data_ptr())
    hip.synchronize(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 8
    ok = int((res[1].cpu().numpy().view(_lib.RESULT_DTYPE)['status'] == 0).sum())
    print(f'{name:30s} S = {loc.template.n_states:3d}/{loc.reverse.n_states:3d}  {hip.kernel_name(0):38s} {hip.kernel_name(1):38s} '
          f'{dt * 1e3:7.2f} ms per call  {n / dt / 1e6:6.3f} M reads/s  called {ok}', flush=True)
    hip.close()
