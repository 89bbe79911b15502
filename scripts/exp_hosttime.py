"""Host-side cost of wsx_call_batch in device mode (time until the asynchronous call returns)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from warpstr_amd import _lib
from warpstr_amd.caller import HipCaller
dev = torch.device('cuda', 0)
n, T = 100000, 2000
locus, signal, offsets, aut = bench.make_workload(n, T, 1000, dev)
hip = HipCaller([locus.template, locus.reverse], [bench.FLANK] * 2, stream=torch.cuda.current_stream().cuda_stream, workspace_limit=96 << 30)
res = torch.zeros((n, 56), dtype=torch.uint8, device=dev)
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hip.call_device(signal.data_ptr(), offsets, aut, res.data_ptr())
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'enqueue {1e3*(t1-t0):.2f} ms, total {1e3*(t2-t0):.2f} ms', flush=True)
