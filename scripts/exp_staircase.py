"""Throughput of the whole call against the number of states: the cost of a 64-state slot boundary.
Usage: exp_staircase.py [n_reads] [samples] [S,S,...]   (device-resident input, pipelined calls, simple repeats with flanks chosen so that the
larger of the two strands' automata has exactly S states)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from warpstr_amd import synth, _lib
from warpstr_amd.caller import HipCaller

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
dev = torch.device('cuda', 0)


def locus_with(S):
    """A locus whose larger automaton has exactly S states (the other strand's at most 4 fewer: the same number of slots
    except right at a boundary)."""
    for pat in ('(AGC)', '(AAAT)', '(CAG)CAACAG(CCG)', '(GGCCCC)'):
        for fl in range(max(12, (S - 30) // 2), (S - 5) // 2 + 4):
            for seed in range(120):
                loc = synth.make_locus(pat, fl, seed)
                a, b = loc.template.n_states, loc.reverse.n_states
                if max(a, b) == S and min(a, b) >= S - 4:
                    return loc, fl, pat
    raise RuntimeError(f'no locus with {S} states')


print(f'{n} reads x {T} samples, both passes, device-resident, 8 pipelined calls', flush=True)
sizes = tuple(int(x) for x in sys.argv[3].split(',')) if len(sys.argv) > 3 else (48, 63, 64, 65, 96, 127, 128, 129, 192, 193, 256, 257, 320)
for S in sizes:
    loc, fl, pat = locus_with(S)
    rng = np.random.default_rng(S)
    base = []
    for _ in range(96):
        rev = bool(rng.random() < 0.5)
        hi = max(1, min(30, (T // 4 - 2 * fl - 12) // 6))
        base.append((synth.squiggle(loc, rev, T, rng, lo=1, hi=hi, sigma=0.0)[0], rev))
    pick = rng.integers(0, len(base), size=n)
    clean = torch.from_numpy(np.stack([b[0] for b in base])).to(dev)
    g = torch.Generator(device=dev); g.manual_seed(S)
    sig = (clean[torch.from_numpy(pick).to(dev)] + 0.25 * torch.randn((n, T), generator=g, device=dev, dtype=torch.float64)).reshape(-1).contiguous()
    aut = np.array([int(base[i][1]) for i in pick], np.int32)
    off = np.arange(n + 1, dtype=np.int64) * T
    res = [torch.zeros((n, 56), dtype=torch.uint8, device=dev) for _ in range(2)]
    hip = HipCaller([loc.template, loc.reverse], [fl, fl], stream=torch.cuda.current_stream().cuda_stream, workspace_limit=64 << 30)
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
    print(f'S = {loc.template.n_states:3d}/{loc.reverse.n_states:3d} {pat} flank {fl:3d}  {hip.kernel_name(0):36s} {dt * 1e3:7.2f} ms per call  {n / dt / 1e6:6.3f} M reads/s  {n * T * S * 2 / dt / 1e12:5.2f} T cells/s  called {ok}', flush=True)
    hip.close()
