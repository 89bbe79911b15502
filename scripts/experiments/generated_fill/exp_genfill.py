"""Fill launches alone (one stream, device-resident headline workload): the generated fill under generator options against the
built-in kernel.  Usage: exp_genfill.py [reads]   (WARPSTR_FILLGEN_OPTS selects the generator variant)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from warpstr_amd import _lib
from warpstr_amd.caller import HipCaller
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dev = torch.device('cuda', 0)
wl = bench.make_headline(n, 2000, 1000, dev)
res = torch.zeros((n, 56), dtype=torch.uint8, device=dev)
for gen in (True, False):
    hip = HipCaller(wl.tables, wl.flanks, stream=torch.cuda.current_stream().cuda_stream, generated_fill=gen)
    hip.set_streams(1)
    for _ in range(2):
        hip.call_device(wl.signal.data_ptr(), wl.offsets, wl.aut, res.data_ptr())
    hip.synchronize()
    hip.call_device(wl.signal.data_ptr(), wl.offsets, wl.aut, res.data_ptr())
    hip.synchronize()
    b, e, r = hip.fill_intervals()
    tm = hip.last_timing()
    print(f"{'generated' if gen else 'built-in '} [{os.environ.get('WARPSTR_FILLGEN_OPTS', '')}] fills (ms, reads): " +
          ', '.join(f'{(y - x):.3f}/{k}' for x, y, k in zip(b, e, r)) + f'  sum {float((e - b).sum()):.3f}  whole call {tm["total_ms"]:.2f} ms', flush=True)
    hip.close()
