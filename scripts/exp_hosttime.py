"""Host-side cost of wsx_call_batch on device buffers (pipelined mode): the time until the asynchronous call returns, against
the time the device needs for it.  Usage: exp_hosttime.py [reads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from warpstr_amd.caller import HipCaller
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dev = torch.device('cuda', 0)
wl = bench.make_headline(n, 2000, 1000, dev)
hip = HipCaller(wl.tables, wl.flanks, stream=torch.cuda.current_stream().cuda_stream)
res = [torch.zeros((n, 56), dtype=torch.uint8, device=dev) for _ in range(4)]
hip.set_pipelined(True)
for k in range(3):
    hip.call_device(wl.signal.data_ptr(), wl.offsets, wl.aut, res[k % 4].data_ptr())
hip.synchronize()
enq = []
t0 = time.perf_counter()
for k in range(20):
    t = time.perf_counter()
    hip.call_device(wl.signal.data_ptr(), wl.offsets, wl.aut, res[k % 4].data_ptr())
    enq.append(time.perf_counter() - t)
hip.synchronize()
dt = time.perf_counter() - t0
# the first two return after their own host work; from the third on a call also waits for the call two before it
print(f'{n} reads: host work of a call {1e3 * min(enq):.2f} ms (first two calls: {1e3 * enq[0]:.2f}, {1e3 * enq[1]:.2f}), step {1e3 * dt / 20:.2f} ms')
