"""Multi-process path on CPU: world_size 2, gloo.  Reads shard with no exchange; one all-gather of the
per-read records; every rank ends up with the full, correctly ordered result table."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp

from warpstr_amd import _lib
from warpstr_amd.dist import gather_results, gather_results_ragged, shard_reads


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_records(idx):
    """Deterministic stand-in for per-read results (the GPU caller is not involved in this CPU test)."""
    rec = np.zeros(len(idx), dtype=_lib.RESULT_DTYPE)
    rec['len2'] = 3 * np.asarray(idx) + 1
    rec['cost2'] = np.asarray(idx) * 0.5
    rec['status'] = np.asarray(idx) % 3 == 0
    return rec


def _worker(rank, world, port, lengths, out_dir):
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    shards = shard_reads(lengths, world)
    mine = shards[rank]
    full = gather_results_ragged(_fake_records(mine), mine, len(lengths), world)
    np.save(os.path.join(out_dir, f'ragged_{rank}.npy'), full)
    # equal-size all-gather used by bench.py
    n = 5
    local = torch.from_numpy(_fake_records(np.arange(rank * n, (rank + 1) * n)).view(np.uint8).reshape(n, -1).copy())
    allr = gather_results(local, world)
    np.save(os.path.join(out_dir, f'equal_{rank}.npy'), allr.numpy())
    dist.destroy_process_group()


class _FakeCaller:
    """Stands in for HipCaller in this CPU test: a deterministic function of each read's samples."""

    def call(self, sig, off, aut):
        rec = np.zeros(len(aut), dtype=_lib.RESULT_DTYPE)
        for i in range(len(aut)):
            s = sig[off[i]:off[i + 1]]
            rec['len2'][i] = len(s) + int(aut[i])
            rec['cost2'][i] = float(s.sum())
        return rec, {}


def _worker_sharded(rank, world, port, out_dir):
    import torch.distributed as dist
    from warpstr_amd.dist import call_sharded
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    rng = np.random.default_rng(5)
    sigs = [rng.normal(size=int(n)) for n in rng.integers(50, 400, size=23)]
    aut = rng.integers(0, 2, size=23)
    full = call_sharded(_FakeCaller(), sigs, aut, world, rank)
    np.save(os.path.join(out_dir, f'sharded_{rank}.npy'), full)
    dist.destroy_process_group()


def test_call_sharded_world2(tmp_path):
    world = 2
    mp.spawn(_worker_sharded, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(5)
    sigs = [rng.normal(size=int(n)) for n in rng.integers(50, 400, size=23)]
    aut = rng.integers(0, 2, size=23)
    for r in range(world):
        got = np.load(os.path.join(tmp_path, f'sharded_{r}.npy'))
        assert [int(v) for v in got['len2']] == [len(s) + int(a) for s, a in zip(sigs, aut)]
        assert np.array_equal(got['cost2'], np.array([float(s.sum()) for s in sigs]))


def test_shard_reads_partition():
    rng = np.random.default_rng(0)
    lengths = rng.integers(500, 5000, size=101)
    for world in (1, 2, 8):
        shards = shard_reads(lengths, world)
        allidx = np.sort(np.concatenate(shards))
        assert np.array_equal(allidx, np.arange(len(lengths)))
        loads = [lengths[s].sum() for s in shards]
        assert max(loads) - min(loads) <= lengths.max()


def test_world2_allgather(tmp_path):
    rng = np.random.default_rng(1)
    lengths = rng.integers(500, 5000, size=37)
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), lengths, str(tmp_path)), nprocs=world, join=True)
    expect = _fake_records(np.arange(len(lengths)))
    for r in range(world):
        got = np.load(os.path.join(tmp_path, f'ragged_{r}.npy'))
        assert got.tobytes() == expect.tobytes()
        eq = np.load(os.path.join(tmp_path, f'equal_{r}.npy')).view(_lib.RESULT_DTYPE).reshape(-1)
        assert np.array_equal(eq['len2'], 3 * np.arange(10) + 1)
