"""How long the reader processes of main_wrapper_loci take to come up (16 x `python -m warpstr_amd._hostworker`), from a process that
has the GPU runtime loaded (as the caller has) and from one that has not, and what a round trip to them costs.
Usage: exp_reader_pool.py [n_workers]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from warpstr_amd.loci import _WorkerPool, _probe_chunk

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16


def trial(tag):
    t0 = time.perf_counter()
    pool = _WorkerPool(n)
    t1 = time.perf_counter()
    pool.procs
    t2 = time.perf_counter()
    pool.map(_probe_chunk, [[] for _ in range(n)])
    t3 = time.perf_counter()
    for _ in range(20):
        pool.map(_probe_chunk, [[] for _ in range(4 * n)])
    t4 = time.perf_counter()
    pool.shutdown()
    print(f'{tag}: constructor {1e3 * (t1 - t0):.1f} ms, all {n} started after {1e3 * (t2 - t0):.1f} ms, first answers after {1e3 * (t3 - t0):.1f} ms; '
          f'a map of {4 * n} empty chunks {1e3 * (t4 - t3) / 20:.2f} ms', flush=True)


trial('plain process')
import torch
torch.zeros(1, device='cuda')
trial('with torch + HIP initialised')
