"""Duration of the fill launches alone (one stream, one chunk) for a bench workload; results are NOT checked (experiment
builds may compute garbage).  Usage: WARPSTR_HIP_LIB=... exp_fill_alone.py <headline|cfg1|cfg5> [reads]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from warpstr_amd import _lib
from warpstr_amd.caller import HipCaller
name = sys.argv[1]; dev = torch.device('cuda', 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else {'headline': 100000, 'cfg1': 20000, 'cfg5': 50000}[name]
if name == 'headline':
    wl = bench.make_headline(n, 2000, 1000, dev)
elif name == 'cfg1':
    pat, fl, tr = bench.CFG1
    wl = bench.make_ragged('cfg1', [(pat, fl, tr, 1, None)], n, 1000, dev)
else:
    wl = bench.make_ragged('cfg5', [(p, bench.cfg5_flank(p, 11 + i), (500, 5000), 11 + i, None) for i, p in enumerate(bench.CFG5_PATTERNS)], n, 1000, dev)
hip = HipCaller(wl.tables, wl.flanks, stream=torch.cuda.current_stream().cuda_stream, workspace_limit=96 << 30)
hip.set_streams(1)
res = torch.zeros((wl.n, 56), dtype=torch.uint8, device=dev)
for rep in range(4):
    hip.call_device(wl.signal.data_ptr(), wl.offsets, wl.aut, res.data_ptr())
    hip.synchronize()
    b, e, r = hip.fill_intervals()
    tm = hip.last_timing()
print(name, os.environ.get('WARPSTR_HIP_LIB', 'default'), hip.kernel_name(0), 'fill launches (ms):', np.round(e - b, 3).tolist(), 'whole call', round(tm['total_ms'], 3))
