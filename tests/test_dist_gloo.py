"""Multi-process path on CPU: world_size 2, gloo.  Reads shard with no exchange; one all-gather of the
per-read records; every rank ends up with the full, correctly ordered result table."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp

from warpstr_amd import _lib
from warpstr_amd.dist import gather_results, gather_results_ragged, shard_reads


def _free_port():
    from tests.helpers import free_port
    return free_port()


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


def test_shard_reads_by_cost_balances_a_mixed_batch():
    """A batch of single-slot and four-slot reads (balance by T x S, not T): with the per-sample cost
    of each read's automaton the modelled loads are within 5 %, by samples alone they are far apart."""
    from warpstr_amd.dist import SLOT_COST, slot_cost
    rng = np.random.default_rng(3)
    lengths = rng.integers(500, 5000, size=4000)
    states = np.where(rng.random(4000) < 0.5, 63, 225)
    cost = np.array([slot_cost(s) for s in states])
    assert slot_cost(63) == SLOT_COST[1] and slot_cost(225) == SLOT_COST[4] and slot_cost(257) == SLOT_COST[5] and slot_cost(700) > SLOT_COST[5]
    for world in (2, 4, 8):
        shards = shard_reads(lengths, world, cost)
        assert np.array_equal(np.sort(np.concatenate(shards)), np.arange(4000))
        loads = np.array([(lengths[s] * cost[s]).sum() for s in shards])
        assert loads.max() / loads.min() <= 1.05
    # what goes wrong without the weights: every single-slot read to one rank, every four-slot read to the other
    order = np.argsort(states, kind='stable')
    lengths, states, cost = lengths[order], states[order], cost[order]
    half = [np.flatnonzero(states == 63), np.flatnonzero(states == 225)]
    assert (lengths[half[1]] * cost[half[1]]).sum() / (lengths[half[0]] * cost[half[0]]).sum() > 2.0


def _fake_called(idx):
    """A CallerResults as CallerWrapper.run_raw returns it for the reads `idx`: per-sample sequence buffers, a failed read."""
    from warpstr_amd.caller import CallerResults
    rec = _fake_records(idx)
    rec['status'] = np.asarray(idx) % 7 == 3
    rec['len1'] = 5 + np.asarray(idx) % 4
    rec['len2'] = 3 + np.asarray(idx) % 5
    offsets = np.arange(len(idx)) * 16
    seq1, seq2 = np.zeros(16 * len(idx) + 1, np.uint8), np.zeros(16 * len(idx) + 1, np.uint8)
    for k, i in enumerate(idx):
        seq1[offsets[k]:offsets[k] + rec['len1'][k]] = np.frombuffer(('ACGT' * 4)[i % 4:][:rec['len1'][k]].encode(), np.uint8)
        seq2[offsets[k]:offsets[k] + rec['len2'][k]] = np.frombuffer(('TTGCA' * 4)[i % 5:][:rec['len2'][k]].encode(), np.uint8)
    return CallerResults([f'r{i}' for i in idx], rec, offsets, seq1, seq2, 'nan')


def _worker_called(rank, world, port, n, out_dir):
    import torch.distributed as dist
    from warpstr_amd.caller import CallerResults
    from warpstr_amd.dist import gather_called
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    shards = shard_reads(100 + 13 * (np.arange(n) % 9), world)
    records, s1, o1, s2, o2 = gather_called(_fake_called(shards[rank]), shards[rank], shards, n, world)
    full = CallerResults([f'r{i}' for i in range(n)], records, o1, s1, s2, 'nan', offsets2=o2)
    with open(os.path.join(out_dir, f'called_{rank}.txt'), 'w') as f:
        f.writelines(f'{i} {int(records["status"][i])} {full[i].seq} {full[i].resc_seq}\n' for i in range(n))
    dist.destroy_process_group()


def test_gather_called_world2(tmp_path):
    """Records AND called sequences of two unequal shards come back complete and in read order on both ranks."""
    n, world = 29, 2
    mp.spawn(_worker_called, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    whole = _fake_called(np.arange(n))
    want = ''.join(f'{i} {int(whole.records["status"][i])} {whole[i].seq} {whole[i].resc_seq}\n' for i in range(n))
    for r in range(world):
        assert open(os.path.join(tmp_path, f'called_{r}.txt')).read() == want
