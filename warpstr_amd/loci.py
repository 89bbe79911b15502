"""Step 3 for ALL loci of a run in one pass over the GPU.

Upstream walks the loci of the configuration one after the other (WarpSTR.py:33-46, 66-76) and calls
`main_wrapper(locus, threads)` for each (src/caller/wrapper.py:17-41): automata, a process pool and a `Pool.map` per
locus.  Real runs have hundreds to thousands of loci with ten to a thousand reads each, so on a GPU a handle, its streams
and its work-set allocations per locus would cost more than the calling.  `main_wrapper_loci` does the same work with ONE
handle: the automata of every locus (template and reverse strand: 2 L of them) are compiled and placed once, the `saved`
reads of all loci form one list that is cut into mixed-locus batches (every read carries its automaton's index), the
batches follow each other on the GPU while the host reads the next one's files, and each locus then gets exactly the
outputs `main_wrapper` writes for it (overview.csv columns, FASTA files, complex-unit table, state_similarity.csv).

Under `python -m torch.distributed.run --nproc-per-node N` (shard=True) the read list is dealt over the N GPUs by cost
(samples x what a sample costs on the read's automaton, dist.slot_cost); every rank reads and calls only its share, two
all-gathers return every record and called sequence to every rank, rank 0 writes.
"""
import os
import time
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from . import overview as ov
from .caller import BatchQueue, CallerConfig, CallerResults, HipCaller, RescalerConfig
from .fast5 import read_raw_signal


class LocusJob:
    """One locus of a multi-locus run: its overview, the rows that are called, its two automata."""

    def __init__(self, locus, pore_model, tm: Dict[str, float]):
        from .automata import locus_automata
        t0 = time.perf_counter()
        self.locus = locus
        self.sequence = locus.sequence.upper()
        self.flank_length = int(locus.flank_length)
        self.overview_path, self.df_overview = ov.load_overview(locus.path)
        df = self.df_overview
        self.saved = np.flatnonzero(np.asarray(df['saved']).astype(bool))
        take = lambda col, dt: np.asarray(df[col])[self.saved].astype(dt)
        self.names = [str(x) for x in df.index.to_numpy()[self.saved]]
        self.reverse = take('reverse', bool)
        self.lo, self.hi = take('l_start_raw', np.int64), take('r_end_raw', np.int64)
        self.run_id = np.asarray(df['run_id'])[self.saved] if 'run_id' in df.columns else None
        self.fast5_path = np.asarray(df['fast5_path'])[self.saved] if 'fast5_path' in df.columns else None
        t1 = time.perf_counter()
        lt, rt, lr, rr = ov.load_flanks(locus.path)
        self.temp_sta, self.rev_sta = locus_automata(lt, rt, lr, rr, self.sequence, pore_model)
        tm['overview_s'] += t1 - t0
        tm['automata_s'] += time.perf_counter() - t1

    @property
    def n(self) -> int:
        return len(self.saved)

    def fast5_of(self, k: int) -> str:
        from .wrapper import annot_fast5_path
        return annot_fast5_path(self.locus.path, self.run_id[k] if self.run_id is not None else 0, self.names[k])

    def raw_read(self, k: int, raw_reader) -> np.ndarray:
        """The whole raw read of saved row k (int16), as get_raw_workload finds it."""
        path = self.fast5_of(k)
        if raw_reader is not read_raw_signal:
            return np.ascontiguousarray(raw_reader(path), dtype=np.int16)
        if not os.path.exists(path) and self.fast5_path is not None:
            return np.ascontiguousarray(read_raw_signal(str(self.fast5_path[k]), self.names[k]), dtype=np.int16)  # caller-only input
        return np.ascontiguousarray(read_raw_signal(path), dtype=np.int16)


class HipEngine:
    """The GPU side of main_wrapper_loci: one handle holding every automaton, on a stream of its own, and the queue of
    batches in flight on it (caller.BatchQueue).  (The CPU tests of the host logic put a stand-in with the same four
    methods in its place; the product has no other engine.)"""

    def __init__(self, tables, flank_lengths, caller_config, rescaler_config, device: int):
        import torch
        self.stream = torch.cuda.Stream(device=torch.device('cuda', device))
        self.hip = HipCaller(tables, flank_lengths, caller_config, rescaler_config, device=device, stream=self.stream.cuda_stream)
        self.queue = BatchQueue(self.hip, self.stream, (caller_config or CallerConfig()).spike_removal)
        self.submit_raw, self.submit_signals, self.collect = self.queue.submit_raw, self.queue.submit_signals, self.queue.collect

    def info(self) -> dict:
        return {'workspace_bytes': self.hip.workspace()['bytes_allocated'], 'workspace_limit_bytes': self.hip.workspace_limit(),
                'kernels': sorted({self.hip.kernel_name(a) for a in range(min(len(self.hip.automata), 256))})}

    def close(self):
        self.hip.synchronize()
        self.hip.close()


def _muted(on: bool):
    """Swallow prints (ranks other than 0; quiet runs)."""
    import contextlib
    import io
    return contextlib.redirect_stdout(io.StringIO()) if on else contextlib.nullcontext()


def _similarity(job: LocusJob, caller_config: CallerConfig, pore_model, write: bool):
    """summaries/state_similarity.csv and upstream's warnings (CallerWrapper.check_high_similarity)."""
    from .caller import CallerWrapper
    cw = CallerWrapper.__new__(CallerWrapper)  # only the similarity report of the class is used: no handle is created
    cw.locus, cw.pore_model, cw.caller_config, cw._write_summaries = job.locus, pore_model, caller_config, write
    cw.check_high_similarity(job.sequence)


# ---- the per-locus host work on several processes (threads > 1) --------------------------------------------------------------
# Upstream's `threads` are Pool workers over the READS of one locus (src/caller/wrapper.py:104-109).  Here the reads are the
# GPU's; what is left on the host is per LOCUS -- parsing its overview, compiling two automata, writing its CSV and FASTA
# files: 4-5 ms of Python and pandas each, thousands of times -- and that is what `threads` spreads: over worker PROCESSES
# (`python -m warpstr_amd._hostworker`: they never share the parent's HIP state and import pandas and this package's host
# modules only).
def _setup_chunk(args):
    loci, caller_config, write, quiet = args
    from .pore_model import default_pore_model
    pm = default_pore_model()
    tm = {'overview_s': 0.0, 'automata_s': 0.0}
    jobs = []
    for locus in loci:
        job = LocusJob(locus, pm, tm)
        with _muted(quiet or not write):
            _similarity(job, caller_config, pm, write=write)
        job.df_overview = None  # (not shipped to the parent and back: whoever writes the locus's files reads it again)
        jobs.append(job)
    return jobs, tm


def _read_chunk(items):
    """Raw reads of (annotated fast5 path, multi-read fall-back path or None, read name) triples, as LocusJob.raw_read finds them."""
    out = []
    for path, fallback, name in items:
        if not os.path.exists(path) and fallback is not None:
            out.append(np.ascontiguousarray(read_raw_signal(fallback, name), dtype=np.int16))
        else:
            out.append(np.ascontiguousarray(read_raw_signal(path), dtype=np.int16))
    return out


def _store_chunk(args):
    from .wrapper import _store_outputs
    out = []
    for (locus, overview_path, df_overview, names, reverse, rec, s1, s2, write, quiet) in args:
        if df_overview is None:
            overview_path, df_overview = ov.load_overview(locus.path)
        l1 = np.where(rec['status'] == 0, rec['len1'], 0).astype(np.int64)
        l2 = np.where(rec['status'] == 0, rec['len2'], 0).astype(np.int64)
        o1, o2 = np.cumsum(l1) - l1, np.cumsum(l2) - l2
        results = CallerResults(names, rec, o1, s1, s2, 'raise', offsets2=o2)
        with _muted(quiet or not write):
            out.append(_store_outputs(locus, overview_path, df_overview, results, [bool(v) for v in reverse], write=write))
    return out


class _WorkerPool:
    """`n` worker processes (`python -m warpstr_amd._hostworker`) and an ordered map over them.  Not multiprocessing's pool:
    its spawned children import the parent's main module again, which a library cannot ask of every script that calls it, and
    forked children would inherit the parent's HIP state."""

    def __init__(self, n: int):
        import subprocess
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get('PYTHONPATH', ''))
        self.procs = [subprocess.Popen([sys.executable, '-m', 'warpstr_amd._hostworker'], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                       env=env) for _ in range(n)]
        self._max_workers = n

    def map(self, func, items):
        """[func(item) for item in items] on the workers (func: a module-level function of this module), in order."""
        import pickle
        import threading
        items = list(items)
        results, errors = [None] * len(items), []
        lock, nxt = threading.Lock(), [0]

        def drive(proc):
            try:
                while not errors:
                    with lock:
                        i = nxt[0]
                        nxt[0] += 1
                    if i >= len(items):
                        return
                    pickle.dump((func.__name__, items[i]), proc.stdin, protocol=pickle.HIGHEST_PROTOCOL)
                    proc.stdin.flush()
                    status, payload = pickle.load(proc.stdout)
                    if status != 'ok':
                        raise RuntimeError(f'{func.__name__} failed in a worker process:\n{payload}')
                    results[i] = payload
            except Exception as e:  # noqa: BLE001 -- raised in the caller's thread below
                errors.append(e)
        threads = [threading.Thread(target=drive, args=(p,), daemon=True) for p in self.procs]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        return results

    def shutdown(self, **_):
        for p in self.procs:
            try:
                p.stdin.close()
            except OSError:
                pass
        for p in self.procs:
            try:
                p.wait(timeout=10)
            except Exception:  # noqa: BLE001
                p.kill()
        self.procs = []


def _pool(threads: int, n_loci: int, n_reads: int = 0):
    """A pool of worker processes for the per-locus host work, or None (one thread; too little work -- fewer than 64 loci and,
    for the output files, fewer than 250 000 reads: starting the workers takes about a second, which the files of 50 000 reads
    do not --; or no way to start one)."""
    if threads <= 1 or (n_loci < 64 and (n_loci < 2 or n_reads < 250000)):
        return None
    try:
        return _WorkerPool(min(int(threads), os.cpu_count() or 1))
    except (ImportError, OSError, ValueError):
        return None


def main_wrapper_loci(loci: Sequence, threads: int = 1, **kwargs):
    """Step 3 for every locus of `loci` through one handle: see _main_wrapper_loci (this wrapper owns the worker processes of
    the per-locus host work, so that they end with the call however it ends)."""
    pools = [_pool(threads, len(loci))]
    try:
        return _main_wrapper_loci(loci, threads, pools, **kwargs)
    finally:
        for pool in pools:
            if pool is not None:
                pool.shutdown()


def _main_wrapper_loci(loci: Sequence, threads: int, pools: list, *, caller_config: Optional[CallerConfig] = None,
                       rescaler_config: Optional[RescalerConfig] = None,
                       signal_loader: Optional[Callable[[str, int, int], np.ndarray]] = None,
                       raw_reader: Callable[[str], np.ndarray] = read_raw_signal, device: int = 0, shard: bool = False,
                       batch_reads: int = 32768, batch_samples: int = 48 << 20, batch_raw_bytes: int = 1 << 30,
                       timings: Optional[Dict[str, float]] = None, quiet: bool = False, _engine=None):
    """Step 3 (src/caller/wrapper.py:17-41) for every locus of `loci` -- objects with `.path`, `.sequence`, `.flank_length`
    (upstream's Locus, src/schemas/locus.py) -- through one handle.  Returns [(df_overview, df_collapsed), ...] in the order of
    `loci` and writes, per locus, exactly what main_wrapper writes.

    signal_loader(fast5path, l_start_raw, r_end_raw) -> normalised float64 segment replaces the GPU loader (default: the int16
    reads go up and are prepared on the device); raw_reader(fast5path) -> int16 read replaces the fast5 reader.
    batch_*: where the read list is cut -- a batch holds at most that many reads, segment samples and raw bytes.
    threads: worker processes for the per-locus host work (overview, automata, output files) from 64 loci on; the reads
    themselves are the GPU's.  shard=True: the run is one torch.distributed job, every rank takes its share of the reads.
    timings: a dict that receives where the wall-clock went (seconds), for the bench's per-locus set-up figure."""
    from . import dist as wdist
    from .pore_model import default_pore_model
    from .wrapper import _store_outputs
    t_start = time.perf_counter()
    tm = timings if timings is not None else {}
    for key in ('overview_s', 'automata_s', 'handle_s', 'read_s', 'submit_s', 'collect_s', 'gather_s', 'store_s'):
        tm[key] = 0.0
    caller_config = caller_config or CallerConfig()
    rank, world = wdist.process_group() if shard else (0, 1)
    collective = shard and (world > 1 or wdist.force_collectives())
    local_gpu = device
    coll_device = None
    if collective:
        import torch
        import torch.distributed as tdist
        local_gpu = int(os.environ.get('LOCAL_RANK', rank)) % max(torch.cuda.device_count(), 1)
        if tdist.get_backend() == 'nccl':
            torch.cuda.set_device(local_gpu)
            coll_device = torch.device('cuda', local_gpu)
    pore_model = default_pore_model()

    # ---- per locus: overview, flanks, automata (every rank: the partition below is derived from them) ---------------
    jobs: List[LocusJob] = []
    pool = pools[0]
    tm['host_processes'] = pool._max_workers if pool is not None else 1
    if pool is not None:
        try:
            step = max(8, min(64, len(loci) // (4 * pool._max_workers) or 8))
            parts = [list(loci[k:k + step]) for k in range(0, len(loci), step)]
            t0 = time.perf_counter()
            for part_jobs, part_tm in pool.map(_setup_chunk, [(p, caller_config, rank == 0, quiet) for p in parts]):
                jobs += part_jobs
                for key, v in part_tm.items():
                    tm[key + '_cpu'] = tm.get(key + '_cpu', 0.0) + v
            tm['overview_s'] = tm['automata_s'] = 0.0
            tm['setup_wall_s'] = time.perf_counter() - t0
        except Exception:  # noqa: BLE001 -- a locus object that does not pickle, a worker that died: do it here
            pool, jobs = None, []
            tm['host_processes'] = 1
    if pool is None:
        for locus in loci:
            job = LocusJob(locus, pore_model, tm)
            with _muted(quiet or rank != 0):  # (upstream prints its similarity warnings once per locus)
                _similarity(job, caller_config, pore_model, write=rank == 0)
            jobs.append(job)
    first = np.zeros(len(jobs) + 1, np.int64)
    np.cumsum([j.n for j in jobs], out=first[1:])
    n_total = int(first[-1])
    tm['n_loci'], tm['n_reads'] = len(jobs), n_total
    if n_total == 0:
        out = [_store_outputs(j.locus, *(ov.load_overview(j.locus.path) if j.df_overview is None else (j.overview_path, j.df_overview)), [], [],
                              write=rank == 0) for j in jobs]
        if collective:
            tdist.barrier()
        tm['total_s'] = time.perf_counter() - t_start
        return out

    # ---- the global read list: (locus, row) -> automaton, cost ----------------------------------------------------------
    locus_of = np.repeat(np.arange(len(jobs)), [j.n for j in jobs])
    row_of = np.concatenate([np.arange(j.n) for j in jobs])
    reverse = np.concatenate([j.reverse for j in jobs])
    lo, hi = np.concatenate([j.lo for j in jobs]), np.concatenate([j.hi for j in jobs])
    aut = (2 * locus_of + reverse).astype(np.int32)
    span = (hi - lo + 1).clip(min=1)
    n_states = np.array([s.n_states for j in jobs for s in (j.temp_sta, j.rev_sta)])
    if collective:
        shards = wdist.shard_reads(span + wdist.READ_OVERHEAD_SAMPLES, world, np.array([wdist.slot_cost(s) for s in n_states])[aut])
    else:
        shards = [np.arange(n_total)]
    mine = shards[rank]

    # ---- rank-local: one handle, mixed-locus batches one behind the other ----------------------------------------------
    error = None
    records = np.zeros(len(mine), dtype=_result_dtype())
    seqs = [[], []]
    try:
        t0 = time.perf_counter()
        queue = (_engine or HipEngine)([s for j in jobs for s in (j.temp_sta, j.rev_sta)], [j.flank_length for j in jobs for _ in range(2)],
                                       caller_config, rescaler_config, local_gpu)
        tm['handle_s'] += time.perf_counter() - t0
        # cut the rank's reads (in global order) into batches
        raw_budget = batch_raw_bytes // 2
        cuts, a, smp = [0], 0, 0
        for k, g in enumerate(mine):
            if k > a and (k - a >= batch_reads or smp + span[g] > batch_samples):
                cuts.append(k)
                a, smp = k, 0
            smp += int(span[g])
        cuts.append(len(mine))
        pending = []  # (ticket, first, count)

        def finish(ticket, b0, b1):
            t1 = time.perf_counter()
            rec, s1, p1, s2, p2 = queue.collect(ticket)
            tm['collect_s'] += time.perf_counter() - t1
            records[b0:b1] = rec
            seqs[0].append(s1)
            seqs[1].append(s2)

        b = 0
        while b < len(cuts) - 1:
            b0, b1 = cuts[b], cuts[b + 1]
            t1 = time.perf_counter()
            data, raw_bytes = [], 0
            if pool is not None and signal_loader is None and raw_reader is read_raw_signal and b1 - b0 >= 64:
                # the fast5 files of a batch on the worker processes (opening a file, HDF5 and the VBZ decoder take about a
                # millisecond per read: in one process more than everything else of a many-loci run together)
                items = []
                for k in range(b0, b1):
                    job, row = jobs[locus_of[mine[k]]], int(row_of[mine[k]])
                    items.append((job.fast5_of(row), str(job.fast5_path[row]) if job.fast5_path is not None else None, job.names[row]))
                step = max(8, min(256, len(items) // (4 * pool._max_workers) or 8))
                for part in pool.map(_read_chunk, [items[k:k + step] for k in range(0, len(items), step)]):
                    data += part
                keep, acc = 0, 0  # long raw reads: as many as fit the byte budget, the rest open the next batch
                while keep < len(data) and (keep == 0 or acc + data[keep].nbytes <= raw_budget):
                    acc += data[keep].nbytes
                    keep += 1
                if keep < len(data):
                    cuts.insert(b + 1, b0 + keep)
                    b1 = b0 + keep
                    del data[keep:]
            else:
                for k in range(b0, b1):
                    g = mine[k]
                    job = jobs[locus_of[g]]
                    if signal_loader is None:
                        data.append(job.raw_read(int(row_of[g]), raw_reader))
                        raw_bytes += data[-1].nbytes
                        if raw_bytes > raw_budget and k + 1 < b1:  # long raw reads: close the batch early
                            cuts.insert(b + 1, k + 1)
                            b1 = k + 1
                            break
                    else:
                        data.append(np.asarray(signal_loader(job.fast5_of(int(row_of[g])), int(lo[g]), int(hi[g])), dtype=np.float64))
            tm['read_s'] += time.perf_counter() - t1
            t1 = time.perf_counter()
            sel = mine[b0:b1]
            if signal_loader is None:
                ticket = queue.submit_raw(data, lo[sel], hi[sel], aut[sel])
            else:
                ticket = queue.submit_signals(data, aut[sel])
            tm['submit_s'] += time.perf_counter() - t1
            pending.append((ticket, b0, b1))
            if len(pending) > 2:  # at most three batches' buffers in HBM / in flight
                finish(*pending.pop(0))
            b += 1
        while pending:
            finish(*pending.pop(0))
        tm.update(queue.info())
        queue.close()
    except Exception as e:  # noqa: BLE001 -- agreed on below: no rank may wait in a collective for one that failed
        error = e
    if collective or error is not None:
        wdist.agree_or_raise(error, world if collective else 1, coll_device, 'reading / calling the reads')

    # ---- the complete table, on every rank ---------------------------------------------------------------------------------
    t0 = time.perf_counter()
    ok = records['status'] == 0
    l1 = np.where(ok, records['len1'], 0).astype(np.int64)
    l2 = np.where(ok, records['len2'], 0).astype(np.int64)
    off1, off2 = np.zeros(len(mine) + 1, np.int64), np.zeros(len(mine) + 1, np.int64)
    np.cumsum(l1, out=off1[1:])
    np.cumsum(l2, out=off2[1:])
    seq1 = np.concatenate(seqs[0]) if seqs[0] else np.zeros(0, np.uint8)
    seq2 = np.concatenate(seqs[1]) if seqs[1] else np.zeros(0, np.uint8)
    if collective:
        local = CallerResults([], records, off1[:-1], seq1, seq2, 'nan', offsets2=off2[:-1])
        records, seq1, off1, seq2, off2 = wdist.gather_called(local, mine, shards, n_total, world, coll_device)
    else:
        off1, off2 = off1[:-1], off2[:-1]
    tm['gather_s'] += time.perf_counter() - t0

    # ---- per locus: the outputs of main_wrapper -------------------------------------------------------------------------------
    t0 = time.perf_counter()
    out = []
    # the first read a caller failed on ends the run where upstream's loop would have ended: the loci before it are written
    bad = np.flatnonzero(records['status'] != 0)
    n_good = int(np.searchsorted(first, bad[0], side='right') - 1) if len(bad) else len(jobs)
    if pool is None:  # few loci with many reads each: the output files are worth a pool of their own
        pool = _pool(threads, n_good, n_total)
        pools.append(pool)
        if pool is not None:
            tm['host_processes'] = pool._max_workers
    if pool is not None and n_good >= 2:
        end1 = np.append(off1, len(seq1)) if len(off1) == n_total else off1
        end2 = np.append(off2, len(seq2)) if len(off2) == n_total else off2
        items = []
        for li, job in enumerate(jobs[:n_good]):
            a, b = int(first[li]), int(first[li + 1])
            items.append((job.locus, job.overview_path, job.df_overview, job.names, job.reverse, records[a:b],
                          seq1[int(end1[a]):int(end1[b])], seq2[int(end2[a]):int(end2[b])], rank == 0, quiet))
        step = max(1, min(64, len(items) // (4 * pool._max_workers) or 1))
        for part in pool.map(_store_chunk, [items[k:k + step] for k in range(0, len(items), step)]):
            out += part
    for li in range(len(out), len(jobs)):
        job = jobs[li]
        if job.df_overview is None:
            job.overview_path, job.df_overview = ov.load_overview(job.locus.path)
        a, b = int(first[li]), int(first[li + 1])
        results = CallerResults(job.names, records[a:b], off1[a:b], seq1, seq2, 'raise', offsets2=off2[a:b]).check()
        with _muted(quiet or rank != 0):
            out.append(_store_outputs(job.locus, job.overview_path, job.df_overview, results, [bool(v) for v in job.reverse],
                                      write=rank == 0))
    tm['store_s'] += time.perf_counter() - t0
    if collective:
        tdist.barrier()  # the files are complete when any rank returns
    tm['total_s'] = time.perf_counter() - t_start
    return out


def _result_dtype():
    from . import _lib
    return _lib.RESULT_DTYPE
