"""Multi-GPU: reads shard embarrassingly across ranks (one process per GPU, no exchange on the data
path); the only collective is one all-gather of the fixed-size per-read result records so that every
rank (or rank 0) can genotype.  Backend 'nccl' is RCCL on ROCm; the same code
runs on 'gloo' with CPU tensors for the world_size-2 tests.
"""
import os
from typing import List, Optional, Sequence

import numpy as np


# What a sample of a read costs, by the number of 64-state slots of its automaton, relative to one slot: the whole call on
# 20 000 reads x 2 000 samples (scripts/exp_staircase.py, profiles/r03_state_staircase.log: 3.11 / 4.8 / 6.3 / 8.0 / 10.4 ms at
# 1..5 slots, the mean over the layouts a slot count can get).  Beyond five slots the general kernel runs (LDS ring, one
# wave per read); its cost per slot was measured once (S = 665: ~6x a register-resident slot).
SLOT_COST = {1: 1.0, 2: 1.54, 3: 2.0, 4: 2.57, 5: 3.35}
# The ratios hold at other read lengths (profiles/r04_staircase_T.log, 20 000 reads per call: 1 : 1.44 : 1.89 : 2.51 at 1 000
# samples, 1 : 1.40 : 1.89 : 2.43 : 3.22 at 3 000, 1 : 1.61 : 2.16 : 2.75 : 3.55 at 5 000), but a read also costs something that
# does not grow with its length (the per-read stages: fit, borders, sort): a one-slot call takes 1.98 / 3.3 / 4.67 / 6.65 ms at
# 1 000 / 2 000 / 3 000 / 5 000 samples -- a + b T with a / b = 700 samples.  Work of a read = (samples + 700) x slot cost.
READ_OVERHEAD_SAMPLES = 700


def slot_cost(n_states: int) -> float:
    """Relative cost per sample of calling a read against an automaton of n_states states (see SLOT_COST)."""
    k = (int(n_states) + 63) // 64
    return SLOT_COST[k] if k in SLOT_COST else 6.0 * k


def _plain_helper():
    """The plain-C loops of warpstr_amd/_seam_helper.so (csrc/seam_helper.c) without the GIL, or None if it is not built."""
    global _PLAIN
    if _PLAIN is False:
        import ctypes as C
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_seam_helper.so')
        _PLAIN = None
        if os.path.exists(path) and not os.environ.get('WARPSTR_NO_SEAM_HELPER'):
            try:
                lib = C.CDLL(path)
                lib.wsx_seam_lpt.restype = None
                lib.wsx_seam_lpt.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
                lib.wsx_seam_gather_pieces.restype = C.c_int64
                lib.wsx_seam_gather_pieces.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
                lib.wsx_seam_scatter_pieces.restype = C.c_int64
                lib.wsx_seam_scatter_pieces.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
                _PLAIN = lib
            except (OSError, AttributeError):
                pass
    return _PLAIN


_PLAIN = False


def _p(a: np.ndarray):
    return a.ctypes.data


def shard_reads(lengths: Sequence[int], world: int, cost_per_sample: Optional[Sequence[float]] = None) -> List[np.ndarray]:
    """Greedy longest-first partition of read indices by work: returns, per rank, the sorted array of read indices it
    owns.  Work = samples x cost_per_sample (slot_cost of the read's automaton: a read on a five-slot automaton costs 3.35 x
    a single-slot one per sample; None = every read costs the same per sample).  Deterministic -- every rank derives the
    same partition from the same description -- and every read appears exactly once.  (The loop over the reads is C when the
    seam helper is built -- 400 000 reads: 0.75 s in Python, a few milliseconds there -- with the same rule and the same
    sums in the same order, so the two give the same partition.)"""
    lengths = np.asarray(lengths, dtype=np.int64)
    work = np.ascontiguousarray(lengths.astype(np.float64) if cost_per_sample is None
                                else lengths * np.asarray(cost_per_sample, dtype=np.float64))
    order = np.ascontiguousarray(np.argsort(-work, kind='stable').astype(np.int64))
    owner = np.empty(len(lengths), dtype=np.int64)
    lib = _plain_helper()
    if lib is not None and world <= 1024:
        lib.wsx_seam_lpt(_p(work), _p(order), len(order), int(world), _p(owner))
    else:
        load = np.zeros(world, dtype=np.float64)
        for i in order:
            r = int(np.argmin(load))
            owner[i] = r
            load[r] += work[i]
    if world == 1:
        return [np.arange(len(lengths))]
    by = np.argsort(owner, kind='stable')
    cuts = np.searchsorted(owner[by], np.arange(world + 1))
    return [by[cuts[r]:cuts[r + 1]] for r in range(world)]


def gather_pieces(src: np.ndarray, starts: np.ndarray, lens: np.ndarray) -> np.ndarray:
    """The ragged pieces src[starts[k] : starts[k] + lens[k]] end to end (uint8)."""
    src = np.ascontiguousarray(src, np.uint8)
    starts, lens = np.ascontiguousarray(starts, np.int64), np.ascontiguousarray(lens, np.int64)
    lib = _plain_helper()
    if lib is None:
        from .caller import ragged_index
        return src[ragged_index(starts, lens)]
    out = np.empty(int(lens.sum()), np.uint8)
    lib.wsx_seam_gather_pieces(_p(src), _p(starts), _p(lens), len(lens), _p(out))
    return out


def scatter_pieces(dst: np.ndarray, src: np.ndarray, idx: np.ndarray, starts: np.ndarray, lens: np.ndarray) -> int:
    """The pieces of src (end to end; the k-th is lens[idx[k]] long) to dst[starts[idx[k]] ...]; returns the bytes consumed."""
    idx = np.ascontiguousarray(idx, np.int64)
    lib = _plain_helper()
    if lib is None:
        from .caller import ragged_index
        n = int(lens[idx].sum())
        dst[ragged_index(starts[idx], lens[idx])] = src[:n]
        return n
    src = np.ascontiguousarray(src, np.uint8)
    return int(lib.wsx_seam_scatter_pieces(_p(src), _p(idx), len(idx), _p(np.ascontiguousarray(starts, np.int64)),
                                           _p(np.ascontiguousarray(lens, np.int64)), _p(dst)))


def force_collectives() -> bool:
    """WARPSTR_DIST_SELF_GATHER=1: a single process still joins a (one-rank) process group and runs every collective -- how
    the RCCL path is exercised on a one-GPU box."""
    return bool(os.environ.get('WARPSTR_DIST_SELF_GATHER'))


def process_group():
    """(rank, world) of this process; joins the job's process group when it was started under torch.distributed.run
    (RANK / WORLD_SIZE / MASTER_* in the environment) and none exists yet.  Backend: WARPSTR_DIST_BACKEND, else 'nccl'
    (= RCCL) when a GPU is visible, else 'gloo'.  The group is created without binding it to a device: an eager
    communicator slowed every later kernel launch on this stack."""
    world_env = int(os.environ.get('WORLD_SIZE', '1'))
    if world_env <= 1 and not force_collectives():
        return 0, 1
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        if world_env <= 1:  # one-rank group outside torch.distributed.run
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            if 'MASTER_PORT' not in os.environ:  # any free port: concurrent one-rank runs on a box must not collide
                import socket
                with socket.socket() as sock:
                    sock.bind(('127.0.0.1', 0))
                    os.environ['MASTER_PORT'] = str(sock.getsockname()[1])
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
        backend = os.environ.get('WARPSTR_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        dist.init_process_group(backend)
    return dist.get_rank(), dist.get_world_size()


def gather_bytes_ragged(local: np.ndarray, world: int, device=None) -> List[np.ndarray]:
    """All-gather one byte string of arbitrary length per rank (the called sequences of a shard): sizes first, then the
    strings padded to the longest.  Returns the per-rank strings."""
    local = np.ascontiguousarray(local, dtype=np.uint8).reshape(-1)
    if world <= 1 and not force_collectives():
        return [local]
    import torch
    import torch.distributed as dist
    dev = device if device is not None else 'cpu'
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, torch.tensor([len(local)], dtype=torch.int64, device=dev))
    sizes = sizes.cpu().numpy()
    cap = max(int(sizes.max()), 1)
    mine = torch.zeros(cap, dtype=torch.uint8, device=dev)
    mine[:len(local)] = torch.from_numpy(local).to(dev)
    everything = torch.empty(world * cap, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(everything, mine)
    everything = everything.cpu().numpy().reshape(world, cap)
    return [everything[r, :int(sizes[r])].copy() for r in range(world)]


def gather_results(local, world: int):
    """All-gather equally sized per-read result records (a [n, 56] uint8 tensor per rank)."""
    if world <= 1:
        return local
    import torch
    import torch.distributed as dist
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return out


def gather_counts(value: int, world: int, device=None) -> np.ndarray:
    """One int64 per rank -> the array of all of them (a fixed-size all-gather: nothing is pickled)."""
    if world <= 1 and not force_collectives():
        return np.array([int(value)], np.int64)
    import torch
    import torch.distributed as dist
    dev = device if device is not None else 'cpu'
    out = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(out, torch.tensor([int(value)], dtype=torch.int64, device=dev))
    return out.cpu().numpy()


def agree_or_raise(error, world: int, device=None, what: str = 'rank-local work'):
    """Every rank calls this after its rank-local part (error: the exception it caught, or None).  If any rank failed,
    every rank raises -- the failing rank its own exception, the others a RuntimeError naming the first failing rank and
    its message -- instead of some ranks waiting in the next collective for one that never comes."""
    if world <= 1 and not force_collectives():
        if error is not None:
            raise error
        return
    msg = (f'{type(error).__name__}: {error}' if error is not None else '').encode('utf-8', 'replace')[:500]
    flags = gather_counts(len(msg), world, device)
    if not flags.any():
        return
    texts = gather_bytes_ragged(np.frombuffer(msg, np.uint8), world, device)
    if error is not None:
        raise error
    first = int(np.flatnonzero(flags)[0])
    raise RuntimeError(f'{what} failed on rank {first}: {texts[first].tobytes().decode("utf-8", "replace")}')


def gather_results_ragged(local_records: np.ndarray, owned: np.ndarray, n_total: int, world: int, device=None):
    """All-gather unequal shards and un-permute to the original read order.
    local_records: structured array (RESULT_DTYPE) of this rank's reads, in the order of `owned`."""
    itemsize = local_records.dtype.itemsize
    if world <= 1 and not force_collectives():
        out = np.zeros(n_total, dtype=local_records.dtype)
        out[owned] = local_records
        return out
    import torch
    import torch.distributed as dist
    counts = gather_counts(len(owned), world, device)
    cap = max(int(counts.max()), 1)
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
        k = int(counts[r])
        idx = np.ascontiguousarray(g[r, :k, itemsize:]).view(np.int64).reshape(-1)
        out[idx] = np.ascontiguousarray(g[r, :k, :itemsize]).view(local_records.dtype).reshape(-1)
    return out


def gather_called(local, owned: np.ndarray, shards: List[np.ndarray], n_total: int, world: int, device=None):
    """Every rank's CallerResults (records + called sequences of its shard, in the order of its `owned` indices) -> the
    complete table in the original read order, on every rank: (records, seq1, offsets1, seq2, offsets2) with the two
    sequence buffers packed (read r's seq at seq1[offsets1[r] : offsets1[r] + len1[r]]).  Two collectives: the 56-byte
    records (+ their read index), and one byte string of sequences per rank.  No per-read Python on either side: the pieces
    move with one fancy-index copy per rank and buffer (400 k reads: a few milliseconds)."""
    records = gather_results_ragged(local.records, owned, n_total, world, device)
    ok = local.records['status'] == 0
    l1 = np.where(ok, local.records['len1'], 0).astype(np.int64)
    l2 = np.where(ok, local.records['len2'], 0).astype(np.int64)
    as_u8 = lambda b: np.frombuffer(b, np.uint8) if isinstance(b, (bytes, bytearray, memoryview)) else np.asarray(b, dtype=np.uint8)
    # a rank's string: the seqs of its reads in shard order, then their resc_seqs
    blob = np.concatenate([gather_pieces(as_u8(local._seq1), local.offsets, l1), gather_pieces(as_u8(local._seq2), local.offsets2, l2)])
    blobs = gather_bytes_ragged(blob, world, device)
    g1 = np.where(records['status'] == 0, records['len1'], 0).astype(np.int64)
    g2 = np.where(records['status'] == 0, records['len2'], 0).astype(np.int64)
    off1, off2 = np.zeros(n_total + 1, np.int64), np.zeros(n_total + 1, np.int64)
    np.cumsum(g1, out=off1[1:])
    np.cumsum(g2, out=off2[1:])
    seq1, seq2 = np.zeros(int(off1[-1]), np.uint8), np.zeros(int(off2[-1]), np.uint8)
    for r in range(world):
        n1 = scatter_pieces(seq1, blobs[r], shards[r], off1, g1)
        scatter_pieces(seq2, blobs[r][n1:], shards[r], off2, g2)
    return records, seq1, off1[:-1], seq2, off2[:-1]


def call_sharded(caller, signals: Sequence[np.ndarray], automaton_id: Sequence[int], world: int, rank: int, device=None,
                 cost_per_sample: Optional[Sequence[float]] = None):
    """Call a whole workload across `world` ranks (one process per GPU): every rank passes the SAME full workload,
    calls only its own shard on its GPU (`caller` = this rank's HipCaller, or anything with a compatible `.call`),
    and receives the complete result table in the original read order (what step 4 / the genotyper consumes).
    The only communication is the all-gather of the 56-byte result records."""
    from .caller import pack_signals
    lengths = [len(s) for s in signals]
    shards = shard_reads(lengths, world, cost_per_sample)
    mine = shards[rank]
    sig, off = pack_signals([signals[i] for i in mine])
    aut = np.asarray(automaton_id, dtype=np.int32)[mine]
    local, _ = caller.call(sig, off, aut)
    return gather_results_ragged(local, mine, len(signals), world, device)
