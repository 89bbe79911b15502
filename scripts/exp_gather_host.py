"""Host time of the multi-rank result gather (dist.gather_called) at configs[4]'s size: 400 000 reads over `world` gloo ranks on
the CPU (the collective itself is RCCL on a node; this measures the Python/numpy part around it).  Usage: exp_gather_host.py [world] [reads]"""
import os
import socket
import sys
import time

import numpy as np
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port, n):
    import torch.distributed as dist
    from warpstr_amd import _lib
    from warpstr_amd.caller import CallerResults
    from warpstr_amd.dist import gather_called, shard_reads
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    rng = np.random.default_rng(1)
    lengths = rng.integers(500, 5000, size=n)
    t0 = time.perf_counter()
    shards = shard_reads(lengths, world)
    t_shard = time.perf_counter() - t0
    mine = shards[rank]
    rec = np.zeros(len(mine), dtype=_lib.RESULT_DTYPE)
    rec['len1'] = 60 + mine % 40
    rec['len2'] = 60 + mine % 37
    rec['status'] = (mine % 1000) == 7
    l1 = np.where(rec['status'] == 0, rec['len1'], 0)
    l2 = np.where(rec['status'] == 0, rec['len2'], 0)
    o1 = np.concatenate([[0], np.cumsum(l1)])[:-1]
    o2 = np.concatenate([[0], np.cumsum(l2)])[:-1]
    s1 = rng.integers(65, 85, size=int(l1.sum()), dtype=np.uint8)
    s2 = rng.integers(65, 85, size=int(l2.sum()), dtype=np.uint8)
    local = CallerResults([], rec, o1, s1, s2, 'nan', offsets2=o2)
    import warpstr_amd.dist as wd
    coll = [0.0]
    real_bytes, real_rec = wd.gather_bytes_ragged, wd.gather_results_ragged

    def timed(f):
        def g(*a, **k):
            t = time.perf_counter()
            out = f(*a, **k)
            coll[0] += time.perf_counter() - t
            return out
        return g
    wd.gather_bytes_ragged, wd.gather_results_ragged = timed(real_bytes), timed(real_rec)
    dist.barrier()
    best, best_coll = 1e9, 0.0
    for _ in range(5):
        coll[0] = 0.0
        t0 = time.perf_counter()
        gather_called(local, mine, shards, n, world)
        dt = time.perf_counter() - t0
        if dt < best:
            best, best_coll = dt, coll[0]
    if rank == 0:
        print(f'world {world}, {n} reads: shard_reads {t_shard * 1e3:.1f} ms, gather_called {best * 1e3:.1f} ms per call, of which the two collectives with their staging (gloo over loopback here; RCCL on a node) {best_coll * 1e3:.1f} ms and the host own index / copy work {(best - best_coll) * 1e3:.1f} ms '
              f'(sequences {int(l1.sum() + l2.sum()) * world / 1e6:.1f} MB in all)')
    dist.destroy_process_group()


if __name__ == '__main__':
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 400000
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(worker, args=(world, port, n), nprocs=world, join=True)
