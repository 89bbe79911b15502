"""Latency of small batches (BASELINE configs[1]: 1k reads x 1.5 kSample, (AGC) flank 16), device-resident inputs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from warpstr_amd import synth, _lib
from warpstr_amd.caller import HipCaller
dev = torch.device('cuda', 0)
locus = synth.make_locus('(AGC)', 16, 42)
for n in (64, 1000, 4000, 16000):
    sigs, revs, _ = synth.batch(locus, min(n, 256), 1500, 42)
    sig = torch.from_numpy(np.concatenate([sigs[i % len(sigs)] for i in range(n)])).to(dev)
    aut = np.array([int(revs[i % len(revs)]) for i in range(n)], np.int32)
    off = np.arange(n + 1, dtype=np.int64) * 1500
    hip = HipCaller([locus.template, locus.reverse], [16, 16], stream=torch.cuda.current_stream().cuda_stream)
    res = torch.zeros((n, 56), dtype=torch.uint8, device=dev)
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hip.call_device(sig.data_ptr(), off, aut, res.data_ptr())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f'n={n}: {dt*1e3:.3f} ms per call -> {n/dt:.3g} reads/s', flush=True)
