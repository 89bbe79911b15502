"""End-to-end cost of the drop-in seam CallerWrapper.run(workload) -> List[CallerResult] (host packing, PCIe, GPU, result
objects) on the headline shape.  Usage: exp_wrapper.py [n_reads]"""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from warpstr_amd import synth
from warpstr_amd.caller import CallerWrapper, ReadSignal
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
locus = synth.make_locus('(AGC)AACAGCCGCCAC(CGC)', 19, 2024, max_states=64)
rng = np.random.default_rng(3)
base = [synth.squiggle(locus, bool(k & 1), 2000, rng, sigma=0.0)[0] for k in range(256)]
work = [ReadSignal(f'r{i}', bool((i % 256) & 1), base[i % 256] + 0.25 * rng.standard_normal(2000)) for i in range(n)]
cw = CallerWrapper.__new__(CallerWrapper)
from warpstr_amd.caller import HipCaller
cw.hip = HipCaller([locus.template, locus.reverse], [19, 19], workspace_limit=64 << 30)
cw.on_error = 'raise'
for rep in range(3):
    t0 = time.perf_counter()
    res = cw.run(work)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    lens = res.lengths()[1]
    strings = [r.resc_seq for r in res]
    dt2 = time.perf_counter() - t1
    print(f'{n} reads: {dt*1e3:.1f} ms per run() -> {n/dt:.3g} reads/s; first: {res[0].resc_seq[:24]}.. cost {res[0].resc_cost:.4f}; '
          f'materialising every CallerResult afterwards: {dt2*1e3:.1f} ms (mean allele length {lens.mean():.1f})', flush=True)
pr = cProfile.Profile(); pr.enable(); cw.run(work); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(12)
