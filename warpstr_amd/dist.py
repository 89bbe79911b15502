"""Multi-GPU: reads shard embarrassingly across ranks (one process per GPU, no exchange on the data
path); the only collective is one all-gather of the fixed-size per-read result records so that every
rank (or rank 0) can genotype.  Backend 'nccl' is RCCL on ROCm; the same code
runs on 'gloo' with CPU tensors for the world_size-2 tests.
"""
from typing import List, Sequence, Tuple

import numpy as np


def shard_reads(lengths: Sequence[int], world: int) -> List[np.ndarray]:
    """Greedy longest-first partition of read indices by work (~ samples): returns, per rank, the sorted
    array of read indices it owns.  Deterministic; every read appears exactly once."""
    lengths = np.asarray(lengths, dtype=np.int64)
    order = np.argsort(-lengths, kind='stable')
    load = np.zeros(world, dtype=np.int64)
    owner = np.empty(len(lengths), dtype=np.int64)
    for i in order:
        r = int(np.argmin(load))
        owner[i] = r
        load[r] += lengths[i]
    return [np.flatnonzero(owner == r) for r in range(world)]


def gather_results(local, world: int):
    """All-gather equally sized per-read result records (a [n, 56] uint8 tensor per rank)."""
    if world <= 1:
        return local
    import torch
    import torch.distributed as dist
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return out


def gather_results_ragged(local_records: np.ndarray, owned: np.ndarray, n_total: int, world: int, device=None):
    """All-gather unequal shards and un-permute to the original read order.
    local_records: structured array (RESULT_DTYPE) of this rank's reads, in the order of `owned`."""
    import torch
    import torch.distributed as dist
    itemsize = local_records.dtype.itemsize
    if world <= 1:
        out = np.zeros(n_total, dtype=local_records.dtype)
        out[owned] = local_records
        return out
    counts = [None] * world
    dist.all_gather_object(counts, int(len(owned)))
    cap = max(counts)
    buf = np.zeros((cap, itemsize + 8), dtype=np.uint8)
    buf[:len(owned), :itemsize] = local_records.view(np.uint8).reshape(len(owned), itemsize)
    buf[:len(owned), itemsize:] = np.asarray(owned, dtype=np.int64).view(np.uint8).reshape(len(owned), 8)
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    g = torch.empty((world * cap, itemsize + 8), dtype=torch.uint8, device=t.device)
    dist.all_gather_into_tensor(g, t)
    g = g.cpu().numpy().reshape(world, cap, itemsize + 8)
    out = np.zeros(n_total, dtype=local_records.dtype)
    for r in range(world):
        k = counts[r]
        idx = np.ascontiguousarray(g[r, :k, itemsize:]).view(np.int64).reshape(-1)
        out[idx] = np.ascontiguousarray(g[r, :k, :itemsize]).view(local_records.dtype).reshape(-1)
    return out


def call_sharded(caller, signals: Sequence[np.ndarray], automaton_id: Sequence[int], world: int, rank: int, device=None):
    """Call a whole workload across `world` ranks (one process per GPU): every rank passes the SAME full workload,
    calls only its own shard on its GPU (`caller` = this rank's HipCaller, or anything with a compatible `.call`),
    and receives the complete result table in the original read order (what step 4 / the genotyper consumes).
    The only communication is the all-gather of the 56-byte result records."""
    from .caller import pack_signals
    lengths = [len(s) for s in signals]
    shards = shard_reads(lengths, world)
    mine = shards[rank]
    sig, off = pack_signals([signals[i] for i in mine])
    aut = np.asarray(automaton_id, dtype=np.int32)[mine]
    local, _ = caller.call(sig, off, aut)
    return gather_results_ragged(local, mine, len(signals), world, device)
