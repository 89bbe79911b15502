"""Soak test of pipelined device-buffer calls: several hundred back-to-back calls of random sizes (1 .. 4 chunk plans, growing
and shrinking work sets), every output compared with the result of a stream-ordered call of the same batch.
Usage: soak_pipelined.py [n_calls]   (test infrastructure; exit code 1 on any difference)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from warpstr_amd import _lib, synth
from warpstr_amd.caller import HipCaller, pack_signals

n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(5)
locus = synth.make_locus('(AGC)AACAGCCGCCAC(CGC)', 19, 2024, max_states=64)
pool, revs, _ = synth.batch(locus, 12000, (500, 1500), 9)
dev = torch.device('cuda:0')
ref = HipCaller([locus.template, locus.reverse], [19, 19])
hip = HipCaller([locus.template, locus.reverse], [19, 19], stream=torch.cuda.current_stream().cuda_stream)
hip.set_pipelined(True)
sizes = [1, 7, 300, 4096, 5000, 8192, 9000, 12000]
cases = []
for n in sizes:
    idx = rng.permutation(len(pool))[:n]
    sig, off = pack_signals([pool[i] for i in idx])
    aut = np.array([1 if revs[i] else 0 for i in idx], dtype=np.int32)
    want, extra = ref.call(sig, off, aut, want_traces=True)
    cases.append((torch.from_numpy(sig).to(dev), off, aut, want, extra['trace2']))
bad = 0
inflight = []
t0 = time.time()
for k in range(n_calls):
    c = int(rng.integers(0, len(cases)))
    dsig, off, aut, want, tr2w = cases[c]
    res = torch.zeros((len(aut), _lib.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    tr2 = torch.zeros(len(dsig), dtype=torch.int16, device=dev)
    hip.call_device(dsig.data_ptr(), off, aut, res.data_ptr(), trace2_ptr=tr2.data_ptr())
    inflight.append((c, res, tr2))
    if len(inflight) >= 6 or k == n_calls - 1:
        hip.join()
        torch.cuda.current_stream().synchronize()
        for c2, r, t in inflight:
            _, _, _, want, tr2w = cases[c2]
            ok = r.cpu().numpy().view(_lib.RESULT_DTYPE).reshape(-1).tobytes() == want.tobytes() and \
                np.array_equal(t.cpu().numpy().view(np.uint16), tr2w)
            bad += 0 if ok else 1
        inflight = []
print(f'{n_calls} pipelined calls, sizes {sizes}: {bad} differing outputs, {time.time() - t0:.1f} s')
sys.exit(1 if bad else 0)
