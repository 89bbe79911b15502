"""Device memory a call allocates under several workspace limits.  Usage: exp_wslimit.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from warpstr_amd import synth
from warpstr_amd.caller import HipCaller, pack_signals
locus = synth.make_locus('(AGC)AACAGCCGCCAC(CGC)', 19, 2024, max_states=64)
sigs, revs, _ = synth.batch(locus, 4000, 1000, 3)
sig, off = pack_signals(sigs)
aut = np.array([1 if x else 0 for x in revs], dtype=np.int32)
dev = torch.device('cuda:0')
dsig = torch.from_numpy(sig).to(dev)
res = torch.zeros((len(aut), 56), dtype=torch.uint8, device=dev)
for limit in (None, 1 << 30, 320 << 20, 160 << 20, 80 << 20):
    torch.cuda.synchronize()
    f0 = torch.cuda.mem_get_info()[0]
    hip = HipCaller([locus.template, locus.reverse], [19, 19], workspace_limit=limit)
    f1 = torch.cuda.mem_get_info()[0]
    hip.call_device(dsig.data_ptr(), off, aut, res.data_ptr())
    hip.synchronize()
    f2 = torch.cuda.mem_get_info()[0]
    print(f'limit {limit}: handle {(f0 - f1) / 2**20:.1f} MiB, call {(f1 - f2) / 2**20:.1f} MiB, fill launches {hip.last_timing()["dp_launches"]}', flush=True)
    hip.close()
