"""Step 3 for ALL loci of a run in one pass over the GPU.

Upstream walks the loci of the configuration one after the other (WarpSTR.py:33-46, 66-76) and calls
`main_wrapper(locus, threads)` for each (src/caller/wrapper.py:17-41): automata, a process pool and a `Pool.map` per
locus.  Real runs have hundreds to thousands of loci with ten to a thousand reads each, so on a GPU a handle, its streams
and its work-set allocations per locus would cost more than the calling.  `main_wrapper_loci` does the same work with ONE
handle: the automata of every locus (template and reverse strand: 2 L of them) are compiled and placed once, the `saved`
reads of all loci form one list that is cut into mixed-locus batches (every read carries its automaton's index), the
batches follow each other on the GPU while the host reads the next one's files, and each locus then gets exactly the
outputs `main_wrapper` writes for it (overview.csv columns, FASTA files, complex-unit table, state_similarity.csv).

The per-locus host work -- overview.csv in and out, two automata, the FASTA files, the complex-unit table -- is native code
without the GIL (csrc/host_loci.cpp through _hostlib; the Python / pandas forms of automata.py, overview.py and units.py are
the definition and the fall-back for whatever the native code declines), so `threads` are THREADS of this process.

Under `python -m torch.distributed.run --nproc-per-node N` (shard=True) the work is dealt over the N GPUs:
  * many loci (>= 8 per rank): by LOCUS -- a rank parses, compiles, reads, calls and WRITES only its own loci (cost of a locus =
    size of its overview.csv x what a sample costs on an automaton of its size); no result travels, one barrier at the end;
  * few loci with many reads (configs[3]): by READ -- every rank sets up every locus, calls its share of the reads (samples x
    dist.slot_cost), two all-gathers return every record and called sequence to every rank, rank 0 writes.
"""
import collections.abc
import os
import sys
import time
from typing import Callable, Dict, List, Mapping, Optional, Sequence

import numpy as np

from . import _hostlib, _readers, overview as ov
from .caller import BatchQueue, CallerConfig, CallerResults, HipCaller, ReadCallError, RescalerConfig, similarity_report
from .fast5 import read_raw_signal

LOCI_PER_RANK_FOR_LOCUS_PARTITION = 8
SHARED_BATCH_BYTES = 320 << 20   # raw bytes of a batch that reader processes decode into a shared staging buffer (80 / 160 / 320 / 640 MB:
                                 # 4.9 / 6.1 / 8.9 / 8.2 k reads/s on 30 000 reads, one box)
SHARED_BATCH_READS = 2048        # ... and its reads


class LocusJob:
    """One locus of a multi-locus run: the `saved` rows of its overview, its two automata, and later its outputs."""

    def __init__(self, locus, pore_model, tm: Dict[str, float], caller_config: Optional[CallerConfig] = None, write: bool = False,
                 native: bool = True, setup=None):
        """setup: the locus's _hostlib.NativeSetup if the caller made it already (a chunk of loci in one library call)."""
        from .automata import locus_automata
        t0 = time.perf_counter()
        self.locus = locus
        self.sequence = locus.sequence.upper()
        self.flank_length = int(locus.flank_length)
        self.overview_path = _hostlib._under(locus.path, ov.OVERVIEW_NAME)
        self._df = None
        lim = caller_config.min_state_similarity if caller_config is not None else 0.0
        # overview.csv, the flank file, both automata and state_similarity.csv in one library call without the GIL; whatever
        # the library leaves out (`None`) is done below by the Python form, which also raises what upstream raises
        if setup is None and native:
            setup = _hostlib.NativeSetup.run(locus.path, self.sequence, pore_model, lim, write and caller_config is not None)
        self._setup = setup
        if self._setup is not None and self._setup.overview_status < 0:
            raise FileNotFoundError(f'Not found the overview file {self.overview_path} - Please check the "output" in config')
        self.native = self._setup.overview if self._setup is not None else None
        if self.native is not None:
            nat = self.native
            self.saved, self.names, self.reverse, self.lo, self.hi = nat.saved, nat.names, nat.reverse, nat.lo, nat.hi
            self.run_id, self.fast5_path = nat.run_id, nat.fast5_path
        else:  # the table goes through pandas, as upstream's does (src/caller/overview.py:37-45)
            self.overview_path, df = ov.load_overview(locus.path)
            self._df = df
            self.saved = np.flatnonzero(np.asarray(df['saved']).astype(bool))
            take = lambda col, dt: np.asarray(df[col])[self.saved].astype(dt)
            self.names = [str(x) for x in df.index.to_numpy()[self.saved]]
            self.reverse = take('reverse', bool)
            self.lo, self.hi = take('l_start_raw', np.int64), take('r_end_raw', np.int64)
            self.run_id = np.asarray(df['run_id'])[self.saved] if 'run_id' in df.columns else None
            self.fast5_path = np.asarray(df['fast5_path'])[self.saved] if 'fast5_path' in df.columns else None
        t1 = time.perf_counter()
        if self._setup is not None and self._setup.tables is not None:
            self.temp_sta, self.rev_sta = self._setup.tables
        else:
            lt, rt, lr, rr = ov.load_flanks(locus.path)
            self.temp_sta, self.rev_sta = locus_automata(lt, rt, lr, rr, self.sequence, pore_model)
        t2 = time.perf_counter()
        # summaries/state_similarity.csv; upstream's warnings are printed by the caller, in the order of the loci
        self.warnings: List[str] = []
        if caller_config is not None:
            if self._setup is not None and self._setup.similarity is not None:
                self.warnings = self._setup.similarity[1]
            else:
                text, self.warnings, _ = similarity_report(self.sequence, pore_model, caller_config.min_state_similarity)
                if write:
                    out_dir = os.path.join(locus.path, 'summaries')
                    os.makedirs(out_dir, exist_ok=True)
                    with open(os.path.join(out_dir, 'state_similarity.csv'), 'w') as f:
                        f.write(text)
        tm['overview_s'] = tm.get('overview_s', 0.0) + t1 - t0   # (native: the whole set-up call is booked here)
        tm['automata_s'] = tm.get('automata_s', 0.0) + t2 - t1
        tm['similarity_s'] = tm.get('similarity_s', 0.0) + time.perf_counter() - t2

    @property
    def n(self) -> int:
        return len(self.saved)

    @property
    def df_overview(self):
        if self._df is None:
            self._df = ov.table_from_text(self.native.text())
        return self._df

    def fast5_of(self, k: int) -> str:
        from .wrapper import annot_fast5_path
        return annot_fast5_path(self.locus.path, self.run_id[k] if self.run_id is not None else 0, self.names[k])

    def rows_for_readers(self, r0: int, r1: int):
        """Saved rows [r0, r1) as ONE item for a reader process (_readers.expand makes the per-read items of it there): the directory
        of the annotated files by run id, the rows' run ids (None: all '0'), names and multi-read fall-backs (None: none).  The
        parent's share of a read is a slice of three lists -- putting 1 700 paths together per batch held the interpreter's lock
        for 7 ms of every 11."""
        pre = self.__dict__.get('_annot_dirs')
        if pre is None:
            from .wrapper import ANNOT_SUBDIR, FAST5_SUBDIR
            runs = set(map(str, self.run_id)) if self.run_id is not None else {'0'}
            pre = self._annot_dirs = {r: os.path.join(self.locus.path, FAST5_SUBDIR, r, ANNOT_SUBDIR) for r in runs}
        return (_readers.ROWS, pre, list(map(str, self.run_id[r0:r1])) if self.run_id is not None else None, list(self.names[r0:r1]),
                list(map(str, self.fast5_path[r0:r1])) if self.fast5_path is not None else None)

    def raw_read(self, k: int, raw_reader) -> np.ndarray:
        """The whole raw read of saved row k (int16), as get_raw_workload finds it."""
        path = self.fast5_of(k)
        if raw_reader is not read_raw_signal:
            return np.ascontiguousarray(raw_reader(path), dtype=np.int16)
        # (files stay open while the run reads them: a multi-read file holds many of a batch's reads; closed by main_wrapper_loci)
        from ._readers import fast5_file, resolve
        path, read_id = resolve((path, str(self.fast5_path[k]) if self.fast5_path is not None else None, self.names[k]))
        return np.ascontiguousarray(fast5_file(path).raw_signal(read_id), dtype=np.int16)


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
        self.stage_shared, self.submit_raw_shared, self.stage_local = self.queue.stage_shared, self.queue.submit_raw_shared, self.queue.stage_local

    # (methods, not attributes: main_wrapper_loci asks the CLASS what an engine can do before it creates one)
    ARENA_REGIONS = BatchQueue.ARENA_REGIONS
    DEVICE_ZSTD = BatchQueue.DEVICE_ZSTD   # (submit_vbz_parts takes chunks whose zstd frame is still around them)

    def submit_raw_parts(self, *a):
        return self.queue.submit_raw_parts(*a)

    def submit_vbz_parts(self, *a):
        return self.queue.submit_vbz_parts(*a)

    def region_wait(self, region):
        return self.queue.region_wait(region)

    def arena_ready(self, path, samples):
        return self.queue.arena_ready(path, samples)

    def arena_is_ready(self, path, samples):
        return self.queue.arena_is_ready(path, samples)


    def add_automata(self, tables, flank_lengths):
        """More loci for the handle while its batches are in flight (wsx_caller_add_automata) -> index of the first new automaton."""
        return self.hip.add_automata(tables, flank_lengths)

    def info(self) -> dict:
        return {'workspace_bytes': self.hip.workspace()['bytes_allocated'], 'workspace_limit_bytes': self.hip.workspace_limit(),
                'handle_create_s': self.hip.create_times(), 'submit_parts_s': dict(self.queue.parts_s), 'collect_parts_s': dict(self.queue.collect_parts),
                'zstd_frames_decoded_on_the_gpu': getattr(self.queue, 'zstd_frames', 0),
                'kernels': sorted({self.hip.kernel_name(a) for a in range(min(len(self.hip.automata), 256))})}

    def close(self):
        self.hip.synchronize()
        self.queue.close()
        self.hip.close()


def _muted(on: bool):
    """Swallow prints (ranks other than 0; quiet runs)."""
    import contextlib
    import io
    return contextlib.redirect_stdout(io.StringIO()) if on else contextlib.nullcontext()


class LociTables(collections.abc.Sequence):
    """What main_wrapper_loci returns: per locus the pair main_wrapper returns, (df_overview, df_collapsed) -- built when it is
    looked at (a run of thousands of loci writes thousands of files and usually looks at none of the tables; a DataFrame costs
    more than the files of its locus).  An entry is ('frames', df_overview, df_collapsed), ('text', overview CSV text, complex-unit
    CSV text or None), ('native', the locus's _hostlib.NativeOverview after its store, complex-unit CSV text or None) or ('disk',
    locus path, whether it may have a complex-unit table: another rank wrote it)."""

    def __init__(self, entries: list):
        self._e = entries

    def __len__(self):
        return len(self._e)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        e = self._e[i]
        if e[0] == 'native':
            e = self._e[i] = ('text', e[1].table_text(), e[2])
        if e[0] == 'text':
            import io

            import pandas as pd
            e = self._e[i] = ('frames', ov.table_from_text(e[1]), None if e[2] is None else pd.read_csv(io.StringIO(e[2]), index_col=0))
        elif e[0] == 'disk':
            import pandas as pd
            cpath = os.path.join(e[1], ov.PREDICTIONS_SUBDIR, ov.COMPLEX_SUBDIR, 'complex_repeat_units.csv')
            e = self._e[i] = ('frames', ov.load_overview(e[1])[1], pd.read_csv(cpath, index_col=0) if e[2] and os.path.exists(cpath) else None)
        return e[1], e[2]


def _complex_header(units, repeat_units):
    """(column names of store_collapsed's table incl. the trailing `reverse`, which generated column each shows) -- the table
    is a dict upstream (overview.py; src/caller/overview.py:11-34): units of the same name share ONE column, at the place of the
    first, with the values of the last --, or None for names a CSV line cannot hold as they are."""
    cols = []
    for unit, alts in zip(units, repeat_units):
        if len(alts) > 1:
            cols.append('main_' + alts[0])
            cols += ['inter_' + a[len(alts[0]):] for a in alts[1:]]
        else:
            cols.append(unit.strip('(').strip(')'))
    last = {}
    for idx, name in enumerate(cols):
        last[name] = idx   # (a dict keeps a key's first position and its last value)
    if 'reverse' in last or any(',' in c or '"' in c or '\n' in c or not c for c in last):
        return None
    return ','.join(list(last) + ['reverse']), list(last.values())


def _store_job(job: LocusJob, rec, seq1, off1, seq2, off2, write: bool, quiet: bool = False, overview_done: bool = False):
    """The outputs of one locus from its reads' records and called sequences (offsets into seq1 / seq2 per read): the files
    main_wrapper writes (write=True) and the entry of LociTables.  overview_done: overview.csv and the FASTA files of this
    (native) locus were written with its chunk (_hostlib.store_many).  Returns (entry, messages to print)."""
    from .wrapper import _store_outputs
    if job.native is not None and (overview_done or bool((rec['status'] == 0).all())):
        l2 = rec['len2']
        if not overview_done:
            job.native.store(job.locus.path, rec['len1'], l2, rec['cost1'], rec['cost2'], seq2, off2, write)
        units, repeat_units, offsets = _units_of(job.sequence)
        if len(units) <= 1:
            return ('native', job.native, None), []
        header = _complex_header(units, repeat_units) if job.n > 0 else None
        got = _hostlib.collapse_store(job.locus.path, seq2, off2, l2, job.reverse, repeat_units, offsets, header[0], header[1], write) if header else None
        if got is not None:
            return ('native', job.native, got[1]), [f'Running complex genotyping as complex repeat units present: {units}']
        # (the complex-unit table through pandas; overview.csv and the FASTA files are written)
        from .units import collapse_repeats
        s2 = bytes(seq2).decode('ascii', 'replace')
        called = [s2[o:o + n] for o, n in zip(np.asarray(off2).tolist(), np.asarray(l2).tolist())]
        df_collapsed = ov.store_collapsed([collapse_repeats(s, repeat_units, offsets) for s in called], units, repeat_units,
                                          [bool(v) for v in job.reverse], job.locus.path, write=write)
        return (('frames', ov.table_from_text(job.native.table_text()), df_collapsed),
                [f'Running complex genotyping as complex repeat units present: {units}'])
    results = CallerResults(job.names, rec, off1, seq1, seq2, 'raise', offsets2=off2).check()
    with _muted(quiet or not write):
        dfo, dfc = _store_outputs(job.locus, job.overview_path, job.df_overview, results, [bool(v) for v in job.reverse], write=write)
    return ('frames', dfo, dfc), []


_UNITS: Dict[str, tuple] = {}


def _units_of(sequence: str):
    """units.break_into_units, remembered per pattern (a run's loci repeat a few hundred patterns at most)."""
    got = _UNITS.get(sequence)
    if got is None:
        from .units import break_into_units
        if len(_UNITS) > 4096:
            _UNITS.clear()
        got = _UNITS[sequence] = break_into_units(sequence)
    return got


# ---- fast5 files on worker processes -----------------------------------------------------------------------------------------
# Opening a file, HDF5 and zstd take a few tenths of a millisecond per read, and libhdf5 is not thread-safe: the one part of the
# host work that runs on worker PROCESSES (`python -m warpstr_amd._hostworker`: they never share the parent's HIP state and
# import the NumPy-free core of the fast5 reader only -- warpstr_amd/_readers.py, _h5core.py).
from ._readers import (decode_arena as _decode_arena, decode_chunk as _decode_chunk, decode_into as _decode_into,  # noqa: E402
                       pack_arena as _pack_arena, probe_chunk as _probe_chunk, read_chunk as _read_chunk)


def _sweep_stale_arenas():
    """Arena files whose reader process no longer exists (a run that was killed before its pool could shut down: the files are
    memory): removed.  A file's name carries its reader's process number."""
    import glob
    import re
    from ._readers import ARENA_DIR
    for path in glob.glob(os.path.join(ARENA_DIR, 'warpstr_arena_*')):
        m = re.match(r'warpstr_arena_(\d+)_', os.path.basename(path))
        if not m:
            continue
        try:
            os.kill(int(m.group(1)), 0)   # (signal 0: does the process exist?)
        except ProcessLookupError:
            try:
                os.unlink(path)
            except OSError:
                pass
        except OSError:   # (exists but is not ours)
            pass


class _WorkerPool:
    """`n` worker processes (`python -m warpstr_amd._hostworker`) and an ordered map over them.  Not multiprocessing's pool:
    its spawned children import the parent's main module again, which a library cannot ask of every script that calls it, and
    forked children would inherit the parent's HIP state."""

    def __init__(self, n: int):
        import subprocess
        import threading
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get('PYTHONPATH', ''))
        self._max_workers = n
        self._procs, self._error, self._drivers = [], None, None

        def start():   # (on a thread of its own: forking a process with the GPU runtime mapped sixteen times takes a good part of a
            try:       # second, in which the caller parses its overviews)
                _sweep_stale_arenas()
                try:   # (the libraries looked up once, here, instead of by every worker)
                    from ._h5core import lib_paths
                    env['WARPSTR_LIBHDF5'], env['WARPSTR_LIBZSTD'] = lib_paths()
                except RuntimeError:
                    pass   # (a worker says which library is missing when it is asked for its first read)
                if not os.environ.get('WARPSTR_NO_READER_ARENAS') and _arena_room(n, 0) is None:
                    # (a forked reader creates its arenas and touches their pages while the loci are still being set up: the first
                    # write into a fresh page of a memory-backed file is a fault, ~50 per read -- a fifth of a reader's time.
                    # Only where the run's room check will also pass: warmed pages stay allocated until the readers end.)
                    env.setdefault('WARPSTR_WARM_ARENAS', str(BatchQueue.ARENA_REGIONS))
                if hasattr(os, 'fork') and not os.environ.get('WARPSTR_NO_FORK_READERS'):
                    self._procs = self._forked(n, env)
                if not self._procs:   # (one interpreter per reader, started from here)
                    for k in range(n):
                        self._procs.append(subprocess.Popen([sys.executable, '-m', 'warpstr_amd._hostworker', str(k + 1)], stdin=subprocess.PIPE,
                                                            stdout=subprocess.PIPE, env=env))
            except OSError as e:
                self._error = e
        self._starter = threading.Thread(target=start, daemon=True)
        self._starter.start()

    @staticmethod
    def _forked(n: int, env):
        """n readers forked by ONE interpreter this process starts (`_hostworker --fork`): starting a process costs this one --
        its address space holds the GPU runtime -- ~10 ms each, and sixteen interpreters importing at once another 80 ms; the
        helper imports once and forks sixteen times in a few milliseconds.  Returns the readers (objects with the pipe ends, the
        process number, wait and kill, as far as the pool uses them of a Popen), or [] if the helper did not come up as expected
        (the caller then starts the readers one by one); raises OSError if it could not be started at all."""
        import subprocess
        parent_ends, child_fds = [], []
        try:
            for _ in range(n):
                task_r, task_w = os.pipe()
                answer_r, answer_w = os.pipe()
                child_fds += [task_r, answer_w]
                parent_ends.append((task_w, answer_r))
            helper = subprocess.Popen([sys.executable, '-m', 'warpstr_amd._hostworker', '--fork'] + [str(fd) for fd in child_fds],
                                      stdout=subprocess.PIPE, env=env, pass_fds=child_fds)
        except OSError:
            for fd in child_fds + [fd for pair in parent_ends for fd in pair]:
                try:
                    os.close(fd)
                except OSError:
                    pass
            raise
        for fd in child_fds:
            os.close(fd)
        import select
        ready, _, _ = select.select([helper.stdout], [], [], 60.0)   # (an interpreter that does not come up is not waited for)
        line = helper.stdout.readline().split() if ready else []
        if len(line) != n or not all(x.isdigit() for x in line):
            helper.kill()
            for fd in [fd for pair in parent_ends for fd in pair]:
                os.close(fd)
            return []

        class Reader:
            def __init__(self, pid, task_w, answer_r):
                self.pid, self.stdin, self.stdout = pid, os.fdopen(task_w, 'wb'), os.fdopen(answer_r, 'rb')

            def wait(self, timeout=None):   # (a reader is the helper's child, not ours: the helper ends when all of them have)
                return helper.wait(timeout=timeout)

            def kill(self):
                import signal
                for victim in (self.pid, helper.pid):
                    try:
                        os.kill(victim, signal.SIGKILL)
                    except OSError:
                        pass
                try:
                    helper.wait(timeout=5)
                except Exception:  # noqa: BLE001
                    pass
        return [Reader(int(pid), tw, ar) for pid, (tw, ar) in zip(line, parent_ends)]

    @property
    def procs(self):
        """The worker processes, all started (raises what starting one of them raised)."""
        if self._starter is not None:
            self._starter.join()
            self._starter = None
        if self._error is not None:
            raise self._error
        return self._procs

    def _drive(self, proc):
        """One thread per worker process: takes the next task of the pool's queue, sends it down the worker's pipe, waits for the
        answer, resolves the task's future.  (The queue is shared: a worker that finishes early takes the next task, whichever
        batch it belongs to.)"""
        import pickle
        while True:
            task = self._tasks.get()
            if task is None:
                return
            name, item, future = task
            if not future.set_running_or_notify_cancel():
                continue
            try:
                pickle.dump((name, item), proc.stdin, protocol=pickle.HIGHEST_PROTOCOL)
                proc.stdin.flush()
                status, payload = pickle.load(proc.stdout)
                if status != 'ok':
                    raise RuntimeError(f'{name} failed in a worker process:\n{payload}')
                future.set_result(payload)
            except BaseException as e:  # noqa: BLE001 -- raised where the future is waited for
                future.set_exception(e)
                if not isinstance(e, RuntimeError):   # the pipe is gone: this worker takes no more tasks
                    return

    def submit(self, func, item):
        """func(item) on a worker (func: a function of warpstr_amd._readers); a concurrent.futures.Future.  An error raised inside
        `func` on the worker is raised by .result() (RuntimeError with the worker's traceback)."""
        import queue as _queue
        import threading
        from concurrent.futures import Future
        if self._drivers is None:
            self._tasks = _queue.Queue()
            self._drivers = [threading.Thread(target=self._drive, args=(p,), name='warpstr-pipe', daemon=True) for p in self.procs]
            for t in self._drivers:
                t.start()
        future = Future()
        self._tasks.put((func.__name__, item, future))
        return future

    def map(self, func, items):
        """[func(item) for item in items] on the workers, in order; the first error is raised."""
        return [f.result() for f in [self.submit(func, item) for item in items]]

    def shutdown(self, **_):
        if self._starter is not None:
            self._starter.join()
            self._starter = None
        if self._drivers is not None:
            for _t in self._drivers:
                self._tasks.put(None)
            for t in self._drivers:
                t.join(timeout=10)
            self._drivers = None
        for p in self._procs:
            try:
                p.stdin.close()
            except OSError:
                pass
        for p in self._procs:
            try:
                p.wait(timeout=10)
            except Exception:  # noqa: BLE001
                p.kill()
        # (a reader removes its arenas when it ends; one that was killed cannot: what carries its process number goes here)
        import glob
        from ._readers import ARENA_DIR
        for p in self._procs:
            for path in glob.glob(os.path.join(ARENA_DIR, f'warpstr_arena_{p.pid}_*')):
                try:
                    os.unlink(path)
                except OSError:
                    pass
        self._procs = []


class _InlinePool:
    """The reader "pool" of a run without reader processes: a chunk is decoded by the thread that hands it out -- the reader
    thread of main_wrapper_loci, beside the calling thread (the library calls inside release the GIL; libhdf5 is used by that one
    thread only).  The same arenas, the same hand-over, one chunk at a time."""
    _max_workers = 1
    inline = True

    def __init__(self):
        self.owner = f'run{id(self):x}.'   # (its regions' keys in _readers: another run on another thread has its own)

    def submit(self, func, item):
        from concurrent.futures import Future
        future = Future()
        try:
            if func in (_decode_arena, _pack_arena):
                item = (f'{self.owner}{item[0]}',) + tuple(item[1:])
            future.set_result(func(item))
        except BaseException as e:  # noqa: BLE001 -- raised where the future is waited for
            future.set_exception(e)
        return future

    def shutdown(self, **_):
        from . import _readers
        _readers._drop_arenas(self.owner)


READER_POOL_FROM_LOCI = 64      # reader processes are started for a run of that many loci ...
READER_POOL_FROM_READS = 2048   # ... or of about that many reads (a few loci with thousands of reads each: configs[3] / [4])


def _reader_pool(threads: int, loci, readers: Optional[int] = None, device_zstd: bool = False):
    """Worker processes for the fast5 files of a run, or None: one thread, or a run too small to be worth sixteen interpreters
    (fewer than READER_POOL_FROM_LOCI loci whose overview.csv files -- ~120 bytes a row -- do not hold READER_POOL_FROM_READS reads
    between them).  Only the START of the processes may fail here (no interpreter, no file descriptors): that is reported once
    and the files are read in this process; what a worker raises while reading is raised by _WorkerPool.map."""
    if threads <= 1 or (readers is not None and readers < 1):
        return None
    if len(loci) < READER_POOL_FROM_LOCI:
        size = 0
        for locus in loci:
            try:
                size += os.stat(os.path.join(locus.path, 'overview.csv')).st_size
            except (OSError, AttributeError):
                pass
        if size < 120 * READER_POOL_FROM_READS:
            return None
    return _WorkerPool(min(int(readers or default_readers(threads, device_zstd)), os.cpu_count() or 1))   # (an explicit `readers` is the caller's word)


def cpu_share() -> int:
    """CPUs this process may actually use: its affinity mask, cut to the cgroup's CPU quota where there is one (a container
    with `cpu.max 1600000 100000` sees 256 CPUs and runs on 16: more busy processes than that are throttled, not scheduled)."""
    try:
        cpus = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cpus = os.cpu_count() or 1
    for path in ('/sys/fs/cgroup/cpu.max', ):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != 'max':
                cpus = min(cpus, max(1, -(-int(quota) // int(period))))
        except (OSError, ValueError):
            pass
    try:   # (cgroup v1)
        quota = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        period = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        if quota > 0 and period > 0:
            cpus = min(cpus, max(1, -(-quota // period)))
    except (OSError, ValueError):
        pass
    return cpus


def default_readers(threads: int, device_zstd: bool = False) -> int:
    """Reader processes of a run with `threads` host threads when the caller does not say: as many as threads (upstream's
    `threads` is its pool size, src/caller/wrapper.py:107-109), never more than the CPUs the process may use (cpu_share) --
    bench.py's from_fast5.reader_sweep on the bench box, whose cgroup grants 16 CPUs: 16 / 32 / 64 / 128 readers = 49 / 55 / 35 /
    22 k reads/s on 60 000 reads (profiles/r06_reader_sweep.json): past the share, more readers only take turns.  device_zstd: the
    engine undoes the chunks' zstd frames itself (wsx_zstd_decode) and a reader's part of a read is libhdf5 alone (0.05 ms instead
    of 0.14); this process's submitting and collecting threads then want a CPU or two of the share for themselves: 8 / 10 / 14 /
    16 readers = 69 / 92 / 89 / 85 k reads/s (medians of three runs each, profiles/r06_reader_sweep_late.json; while the uploads
    still waited for the kernels of the batch before and the readers' items were put together in this process, eight readers were
    the knee: profiles/r06_reader_sweep_device_zstd.json)."""
    n = max(1, min(int(threads), cpu_share()))
    return n - max(1, n // 8) if device_zstd and n >= 4 else n


def _started(pool, tm):
    """The pool once its processes are up, or None (reported once) if they could not be started."""
    if pool is None:
        return None
    try:
        pool.procs
        return pool
    except OSError as e:
        print(f'warpstr_amd: could not start {pool._max_workers} reader processes ({e}); reading the fast5 files in this process', file=sys.stderr)
        tm['reader_processes'] = 0
        return None


_SPREAD = [0]


def spread_over_cpus(k: Optional[int] = None):
    """Move the calling thread to the k-th CPU of the process's affinity mask and release it again (the mask is restored at
    once: nothing stays pinned).  A new thread starts on its parent's CPU and is moved by the kernel's load balancer, which on
    virtualised hosts takes longer than a phase of this driver lasts: sixteen threads created for 100 ms of work then share one
    CPU (8 threads: 77 k automata/s spread, 9 k/s not: scripts/exp_cpu_parallel.py).  k=None: the next index of a process-wide
    counter, offset by LOCAL_RANK so that the ranks of a node do not start on the same CPUs."""
    if not hasattr(os, 'sched_setaffinity'):
        return
    try:
        allowed = sorted(os.sched_getaffinity(0))
        if len(allowed) < 2:
            return
        if k is None:
            k = _SPREAD[0]
            _SPREAD[0] += 1
        k += int(os.environ.get('LOCAL_RANK', '0') or 0) * 16
        os.sched_setaffinity(0, {allowed[k % len(allowed)]})
        os.sched_setaffinity(0, allowed)
    except OSError:
        pass


def _thread_map(executor, func, items):
    """[func(x) for x in items], on the executor's threads if there is one; the first exception is raised here."""
    if executor is None:
        return [func(x) for x in items]
    return list(executor.map(func, items))


def partition_loci(loci: Sequence, world: int) -> List[np.ndarray]:
    """Which rank owns which locus (sorted index arrays per rank), derived alike on every rank without opening a table: cost of
    a locus = bytes of its overview.csv (proportional to its reads) x the cost of a sample on an automaton of about its size
    (dist.slot_cost of 2 x flank + pattern length), dealt greedily, most expensive first (dist.shard_reads)."""
    from . import dist as wdist
    cost = np.ones(len(loci), np.int64)
    per_sample = np.ones(len(loci), np.float64)
    for i, locus in enumerate(loci):
        try:
            cost[i] = max(os.path.getsize(os.path.join(locus.path, ov.OVERVIEW_NAME)), 1)
        except OSError:
            pass  # (the rank that owns it raises upstream's error for the missing table)
        per_sample[i] = wdist.slot_cost(2 * int(locus.flank_length) + len(locus.sequence))
    return wdist.shard_reads(cost, world, per_sample)


GC_PAUSE_FROM_LOCI = 64
ARENA_ROOM_PER_READ = 400 << 10   # bytes of /dev/shm a read may take in a reader arena (a 200 k-sample read as int16 samples)
ARENA_MIN_BYTES = 16 << 20        # a reader's arena is at least this big (_readers._arena)
ARENA_PROBE_READS = 64            # reads of an arena batch while the run has decoded none yet (their mean size then sizes the batches)


def _arena_room(workers: int, reads: int, byte_budget: int = 0) -> Optional[str]:
    """None if /dev/shm has room for the reader arenas of a run -- BatchQueue.ARENA_REGIONS regions of a batch each
    (ARENA_ROOM_PER_READ a read, or the batch's byte budget if that is smaller; ARENA_MIN_BYTES a reader process at least) and
    half as much again --, else the sentence that says what is missing.  A container's default /dev/shm of 64 MB has not: the
    run then takes the staging ring (which falls back to the pipes by itself) or, in one process, the page-locked ring."""
    if not os.path.isdir('/dev/shm'):
        return '/dev/shm does not exist'
    per_batch = reads * ARENA_ROOM_PER_READ
    if byte_budget:
        per_batch = min(per_batch, 2 * byte_budget)   # (a batch is cut at its byte budget: arena_batches)
    need = int(1.5 * BatchQueue.ARENA_REGIONS * max(workers * ARENA_MIN_BYTES, per_batch))
    st = os.statvfs('/dev/shm')
    if st.f_bavail * st.f_frsize < need + (64 << 20):
        return f'/dev/shm has {st.f_bavail * st.f_frsize >> 20} MB free, the reader arenas may take {need >> 20} MB'
    return None


CHUNKS_PER_READER = int(os.environ.get('WARPSTR_CHUNKS_PER_READER', '2') or 2)   # chunks a batch gives every reader process (a round trip costs ~0.1 ms)
STREAM_FROM_LOCI = 256   # a run of that many loci (one rank, fast5 files, reader arenas) reads its first loci's files while it still
                         # sets the later ones up


def _mark(tm, name):
    """A point of a run's timeline -- seconds since main_wrapper_loci was called, and the CPU seconds this process has used since --
    kept when the caller's `timings` dict has a 'timeline' list (WARPSTR_BENCH_TIMELINE=1 scripts/exp_from_fast5.py)."""
    tl = tm.get('timeline')
    if tl is not None:
        tl.append((f'{name} [cpu {time.process_time() - tm["_c0"]:.3f}]', round(time.perf_counter() - tm['_t0'], 4)))


def _streamed_run(parts, setup, tm, pool, engine_cls, engine_args, batch_reads, batch_samples, raw_budget, gpu_vbz, print_warnings, gpu_zstd=False):
    """Set-up, reading and calling of a run as ONE pipeline (upstream's loop reaches a locus, builds its automata, calls its
    reads: WarpSTR.py:33-76): the loci are set up part after part on a thread of its own; as soon as the first part is there the
    reader thread hands its reads' files to the reader processes; the calling thread creates the handle from the loci known by
    then and ADDS the later ones as they come (wsx_caller_add_automata), submitting every batch as it is decoded.  The readers
    no longer idle while thousands of loci are parsed and compiled, nor the set-up threads while the files are read.
    Reader arenas only (loci.main_wrapper_loci decides).  (Reads that are in host memory already -- raw_reads -- gain nothing from
    it: every phase of such a run is this process's own work, and threads of one interpreter take turns; measured on 2 000 loci x 30
    reads, same box: 6.5 k loci/s streamed, 6.7-7.6 k set up first, profiles/r06_streamed_many_loci_ab.json.)  Returns (jobs, first read of every job, records, [seq1 parts], [seq2 parts])."""
    import collections
    import queue as _queue
    import threading
    cond = threading.Condition()
    jobs: List[LocusJob] = []

    class Known:   # what is known of the run's reads so far (grown under `cond`; a prefix, once handed out, never changes)
        n = 0
        cap = 0
        lo = hi = span = least = np.zeros(0, np.int64)
        aut = np.zeros(0, np.int32)
        locus = row = np.zeros(0, np.int64)
        first = [0]
        final = False
        error: Optional[BaseException] = None
    K = Known

    def grow(part_jobs):
        counts = [j.n for j in part_jobs]
        m = int(sum(counts))
        base_job = len(jobs)
        with cond:
            if K.n + m > K.cap:
                K.cap = max(2 * K.cap, K.n + m, 4096)
                for name in ('lo', 'hi', 'span', 'least', 'aut', 'locus', 'row'):
                    old = getattr(K, name)
                    new = np.zeros(K.cap, old.dtype)
                    new[:K.n] = old[:K.n]
                    setattr(K, name, new)
            a, b = K.n, K.n + m
            if m:
                lo = np.concatenate([j.lo for j in part_jobs])
                hi = np.concatenate([j.hi for j in part_jobs])
                loc = np.repeat(np.arange(base_job, base_job + len(part_jobs)), counts)
                K.lo[a:b], K.hi[a:b] = lo, hi
                K.span[a:b] = (hi - lo + 1).clip(min=1)
                K.least[a:b] = 2 * (hi + 1).clip(min=1)
                K.locus[a:b] = loc
                K.row[a:b] = np.concatenate([np.arange(c) for c in counts])
                K.aut[a:b] = (2 * loc + np.concatenate([j.reverse for j in part_jobs])).astype(np.int32)
            for c in counts:
                K.first.append(K.first[-1] + c)
            jobs.extend(part_jobs)
            K.n = b
            cond.notify_all()

    t_setup = time.perf_counter()
    _mark(tm, 'the streamed run begins')

    def run_setup():
        try:
            spread_over_cpus()
            for part in parts:
                if stop.is_set():   # (the run has failed elsewhere: nothing more to set up)
                    break
                part_jobs, ptm = setup(part)
                for key, v in ptm.items():
                    tm[key] = tm.get(key, 0.0) + v
                print_warnings(part_jobs)
                grow(part_jobs)
                _mark(tm, 'part set up')
        except BaseException as e:  # noqa: BLE001 -- raised by the calling thread
            K.error = e
        finally:
            tm['setup_wall_s'] = time.perf_counter() - t_setup
            _mark(tm, 'set-up done')
            with cond:
                K.final = True
                cond.notify_all()

    stop, engine_ready = threading.Event(), threading.Event()
    submitted: Dict[int, threading.Event] = {}
    handover: '_queue.Queue' = _queue.Queue(maxsize=1)
    engine = [None]
    regions = getattr(engine_cls, 'ARENA_REGIONS', 3)
    inline = getattr(pool, 'inline', False)
    reads_cap = min(batch_reads, SHARED_BATCH_READS // 4 if inline else SHARED_BATCH_READS)

    def item_of(k):
        job, row = jobs[int(K.locus[k])], int(K.row[k])
        return (job.fast5_of(row), str(job.fast5_path[row]) if job.fast5_path is not None else None, job.names[row])

    def rows_of(x0, x1):
        """Reads [x0, x1) of the run as the readers take them: a piece per locus (LocusJob.rows_for_readers)."""
        out, j = [], int(K.locus[x0])
        while x0 < x1:
            job, r0 = jobs[j], x0 - K.first[j]
            r1 = min(job.n, r0 + x1 - x0)
            if r1 > r0:
                out.append(job.rows_for_readers(r0, r1))
            x0, j = x0 + r1 - r0, j + 1
        return out

    def page_lock(parts_):
        if not engine_ready.is_set() or engine[0] is None:
            return False
        ready = getattr(engine[0], 'arena_ready', None)
        if ready is not None:
            t1 = time.perf_counter()
            for part in parts_:
                ready(part[0], part[1] // 2 if len(part) == 6 else part[1])
            if time.perf_counter() - t1 > 2e-4:   # (an arena that was not page-locked yet)
                tm['page_lock_s'] = tm.get('page_lock_s', 0.0) + time.perf_counter() - t1
        return True

    def lock_early(future):
        """A chunk's answer has come (this runs on the thread that serves its reader): the arena it names is page-locked now, beside
        the other readers' -- not one after the other when the batch's last answer is in (2 ms each: the run's first three batches,
        whose arenas are all new, each waited 30-60 ms for that)."""
        try:
            if future.exception() is None:
                got = future.result()
                page_lock([(got[0], got[1], None, None, None, None) if gpu_vbz else (got[0], got[1])])
        except Exception:  # noqa: BLE001 -- best effort: arena_batches page-locks what is not yet
            pass

    def lock_all(parts_):
        """What lock_early has left (answers that came before the handle existed): the arenas side by side, not one after the other."""
        known = getattr(engine[0], 'arena_is_ready', None) if engine_ready.is_set() and engine[0] is not None else None
        if known is not None:   # (after a run's first batches every arena is: nothing to start threads for)
            todo = [part for part in {part[0]: part for part in parts_}.values() if not known(part[0], part[1] // 2 if len(part) == 6 else part[1])]
            threads_ = [threading.Thread(target=page_lock, args=([part],), daemon=True) for part in todo[1:]]
            for t in threads_:
                t.start()
            page_lock(todo[:1])
            for t in threads_:
                t.join()
        else:
            page_lock(parts_)

    def arena_batches():
        inflight = collections.deque()
        b, k = 0, 0
        seen = [0, 0]
        while True:
            while len(inflight) < regions - 1:
                with cond:
                    n, final = K.n, K.final
                    span, least = K.span, K.least
                if b >= n:
                    break
                if not final and n - b < reads_cap // 8 and inflight:
                    break   # (a sliver of a part: more is on its way, and the readers have work)
                b1 = min(n, b + reads_cap)
                if not seen[0]:
                    b1 = min(b1, b + ARENA_PROBE_READS)
                mean = seen[1] // seen[0] if seen[0] else 0
                fit = min(int(np.searchsorted(np.cumsum(np.maximum(least[b:b1], mean)), raw_budget, side='right')),
                          int(np.searchsorted(np.cumsum(span[b:b1]), batch_samples, side='right')))
                b1 = b + max(1, min(b1 - b, fit))
                t1 = time.perf_counter()
                region = k % regions
                if k >= regions:
                    while not submitted[k - regions].wait(0.2):
                        if stop.is_set():
                            return
                    if stop.is_set():
                        return
                    engine[0].region_wait(region)
                    del submitted[k - regions]
                submitted[k] = threading.Event()
                step = max(8, -(-(b1 - b) // (CHUNKS_PER_READER * pool._max_workers)))
                futures = []
                for q in range(b, b1, step):
                    items = rows_of(q, min(q + step, b1))
                    futures.append(pool.submit(_pack_arena if gpu_vbz else _decode_arena, (region, k, items, gpu_zstd) if gpu_vbz else (region, k, items)))
                    futures[-1].add_done_callback(lock_early)
                inflight.append((b, b1, region, futures, k))
                tm['read_s'] += time.perf_counter() - t1
                _mark(tm, f'batch {k} handed to the readers ({b1 - b} reads)')
                b, k = b1, k + 1
            if not inflight:
                with cond:
                    if K.n > b:
                        continue
                    if K.final:
                        return
                    cond.wait(0.2)
                if stop.is_set():
                    return
                continue
            b0, b1, region, futures, kb = inflight.popleft()
            t1 = time.perf_counter()
            parts_ = []
            try:
                answers = [f.result() for f in futures]
            except RuntimeError as e:
                if 'no room for a reader arena' not in str(e):
                    raise
                if not tm.get('arena_fallbacks'):
                    print(f'warpstr_amd: {str(e).strip().splitlines()[-1]}; such batches are read without arenas', file=sys.stderr)
                tm['arena_fallbacks'] = tm.get('arena_fallbacks', 0) + 1
                for f in futures:
                    f.exception()
                items = [item_of(x) for x in range(b0, b1)]
                data = []
                if inline:
                    data = _read_chunk(items)
                else:
                    step = max(8, -(-len(items) // (CHUNKS_PER_READER * pool._max_workers)))
                    for part in pool.map(_read_chunk, [items[q:q + step] for q in range(0, len(items), step)]):
                        data += part
                seen[0] += b1 - b0
                seen[1] += int(sum(d.nbytes for d in data))
                tm['raw_bytes'] = tm.get('raw_bytes', 0) + int(sum(d.nbytes for d in data))
                submitted[kb].set()
                tm['read_s'] += time.perf_counter() - t1
                yield b0, b1, data, None
                continue
            seen[0] += b1 - b0
            for got in answers:
                if gpu_vbz:
                    path, cap, base, used, lens_p, table, busy = got
                    parts_.append((path, cap, base, used, lens_p, table))
                    tm['uploaded_bytes'] = tm.get('uploaded_bytes', 0) + int(used)
                else:
                    path, cap, base, lens_p, busy = got
                    parts_.append((path, cap, base, lens_p))
                    tm['uploaded_bytes'] = tm.get('uploaded_bytes', 0) + 2 * int(sum(lens_p))
                tm['decode_worker_s'] = tm.get('decode_worker_s', 0.0) + float(busy)
                tm['raw_bytes'] = tm.get('raw_bytes', 0) + 2 * int(sum(lens_p))
                seen[1] += 2 * int(sum(lens_p))
            lock_all(parts_)
            tm['decode_s'] = tm.get('decode_s', 0.0) + time.perf_counter() - t1
            tm['read_s'] += time.perf_counter() - t1
            if tm.get('timeline') is not None:
                busy_all = [float(got[-1]) for got in answers]
                _mark(tm, f'batch {kb} answered (its {len(answers)} chunks took the readers {sum(busy_all) * 1e3:.0f} ms, the slowest {max(busy_all) * 1e3:.1f})')
            yield b0, b1, parts_, ('vbz' if gpu_vbz else 'arena', region, kb)

    def produce():
        try:
            for item in arena_batches():
                locked = item[3] is None
                while not stop.is_set():
                    locked = locked or (engine_ready.is_set() and engine[0] is not None and (lock_all(item[2]) or True))
                    try:
                        handover.put(item, timeout=0.2 if locked else 0.01)
                        break
                    except _queue.Full:
                        pass
                if stop.is_set():
                    return
            item = None
        except BaseException as e:  # noqa: BLE001 -- raised by the consumer below
            item = e
        while not stop.is_set():
            try:
                handover.put(item, timeout=0.2)
                return
            except _queue.Full:
                pass

    held = [0]   # jobs whose automata the handle holds

    def ensure_automata(n_jobs_needed):
        """The handle holds the automata of every job known by now (at least the first n_jobs_needed)."""
        with cond:
            n_jobs = len(jobs)
        assert n_jobs >= n_jobs_needed
        if n_jobs == held[0]:
            return
        t0 = time.perf_counter()
        new = jobs[held[0]:n_jobs]
        tables = [s for j in new for s in (j.temp_sta, j.rev_sta)]
        flanks = [j.flank_length for j in new for _ in range(2)]
        if engine[0] is None:
            engine[0] = engine_cls(tables, flanks, *engine_args)
            engine_ready.set()
        else:
            first_new = engine[0].add_automata(tables, flanks)
            if first_new != 2 * held[0]:
                raise RuntimeError(f'the handle numbered the new automata from {first_new}, the run from {2 * held[0]}')
            tm['automata_added_in_flight'] = tm.get('automata_added_in_flight', 0) + len(tables)
        held[0] = n_jobs
        tm['handle_s'] += time.perf_counter() - t0
        _mark(tm, f'handle holds {n_jobs} loci')

    rec_parts, seqs = [], [[], []]

    def finish(ticket, b0, b1):
        t1 = time.perf_counter()
        rec, s1, p1, s2, p2 = engine[0].collect(ticket)
        tm['collect_s'] += time.perf_counter() - t1
        _mark(tm, f'reads {b0}-{b1} collected')
        rec_parts.append((b0, b1, rec))
        seqs[0].append(s1)
        seqs[1].append(s2)

    def submit(b0, b1, data, slot):
        with cond:
            lo, hi, aut, locus = K.lo, K.hi, K.aut, K.locus
        ensure_automata(int(locus[b1 - 1]) + 1)
        t1 = time.perf_counter()
        if slot is None:
            ticket = engine[0].submit_raw(data, lo[b0:b1], hi[b0:b1], aut[b0:b1])
        elif slot[0] == 'vbz':
            ticket = engine[0].submit_vbz_parts(slot[1], data, lo[b0:b1], hi[b0:b1], aut[b0:b1])
        else:
            ticket = engine[0].submit_raw_parts(slot[1], data, lo[b0:b1], hi[b0:b1], aut[b0:b1])
        tm['submit_s'] += time.perf_counter() - t1
        _mark(tm, f'reads {b0}-{b1} submitted')
        if slot is not None:
            submitted[slot[2]].set()
        collect_q.put((ticket, b0, b1))   # (three batches may wait there: the submitting thread then waits for the collecting one)
        if collector_error:
            raise collector_error[0]

    # The batches' results are fetched on a thread of their own: packing a batch's sequences is a dozen small launches that queue
    # behind the next batch's kernels (3 ms a batch, none of it this process's work) -- the thread that submits does not wait for it.
    collect_q: '_queue.Queue' = _queue.Queue(maxsize=3)
    collector_error: List[BaseException] = []

    def run_collector():
        while True:
            got = collect_q.get()
            if got is None:
                return
            if collector_error:
                continue   # (the run has failed: what is still queued is dropped, the submitting thread raises)
            try:
                finish(*got)
            except BaseException as e:  # noqa: BLE001 -- raised by the submitting thread
                collector_error.append(e)
                stop.set()

    setup_thread = threading.Thread(target=run_setup, name='warpstr-setup', daemon=True)
    reader = threading.Thread(target=produce, name='warpstr-reader', daemon=True)
    collector = threading.Thread(target=run_collector, name='warpstr-collector', daemon=True)
    setup_thread.start()
    collector.start()
    try:
        try:
            reader.start()
            while True:
                try:
                    item = handover.get(timeout=0.2)
                except _queue.Empty:
                    if collector_error:   # (the reader thread has stopped with the run: nothing more will come)
                        raise collector_error[0]
                    continue
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                submit(*item)
        finally:
            collect_q.put(None)
            collector.join()
        if collector_error:
            raise collector_error[0]
    finally:
        stop.set()
        engine_ready.set()
        _mark(tm, 'last batch collected')
        reader.join()
        setup_thread.join()
        if engine[0] is not None:
            try:
                tm.update(engine[0].info())
            finally:
                engine[0].close()
                _mark(tm, 'handle closed')
    if K.error is not None:
        raise K.error
    n_total = K.n
    records = np.zeros(n_total, dtype=_result_dtype())
    for b0, b1, rec in rec_parts:
        records[b0:b1] = rec
    return jobs, np.asarray(K.first, np.int64), records, seqs


def main_wrapper_loci(loci: Sequence, threads: int = 1, *, caller_config: Optional[CallerConfig] = None,
                      rescaler_config: Optional[RescalerConfig] = None,
                      signal_loader: Optional[Callable[[str, int, int], np.ndarray]] = None,
                      raw_reader: Callable[[str], np.ndarray] = read_raw_signal,
                      raw_reads: Optional[Mapping[str, np.ndarray]] = None, pore_model=None, device: int = 0, shard: bool = False,
                      readers: Optional[int] = None, partition: str = 'auto', batch_reads: int = 32768, batch_samples: int = 48 << 20, batch_raw_bytes: int = 1 << 30,
                      timings: Optional[Dict[str, float]] = None, quiet: bool = False, native: bool = True, _engine=None):
    """Step 3 (src/caller/wrapper.py:17-41) for every locus of `loci` -- objects with `.path`, `.sequence`, `.flank_length`
    (upstream's Locus, src/schemas/locus.py) -- through one handle.  Returns a LociTables: [(df_overview, df_collapsed), ...] in
    the order of `loci` (built when looked at), and writes, per locus, exactly what main_wrapper writes.

    signal_loader(fast5path, l_start_raw, r_end_raw) -> normalised float64 segment replaces the GPU loader (default: the int16
    reads go up and are prepared on the device); raw_reader(fast5path) -> int16 read replaces the fast5 reader; raw_reads: a
    mapping read name -> int16 read for reads that are in host memory already (no per-read call-back).
    pore_model: a pore_model.PoreModel (default: the built-in r9.4 table; `pore_model_path` of a configuration).
    batch_*: where the read list is cut -- a batch holds at most that many reads, segment samples and raw bytes.
    threads: threads of the per-locus host work (native code without the GIL: overview, automata, output files); with more than
    one thread and 64 loci or more (or fewer loci whose overviews hold ~2 000 reads or more) the fast5 files are read on worker
    processes: `readers` of them (default: default_readers(threads); 0: none).
    shard=True: the run is one torch.distributed job.  partition: 'loci' = every rank takes whole loci, 'reads' = every rank
    takes its share of every locus's reads, 'auto' = 'loci' from LOCI_PER_RANK_FOR_LOCUS_PARTITION loci per rank on.
    timings: a dict that receives where the wall-clock went (seconds), for the bench's per-locus set-up figure.

    Python's cyclic garbage collector is paused for the duration of a run of GC_PAUSE_FROM_LOCI loci or more (and switched on again
    when the call returns or raises; WARPSTR_KEEP_GC=1 leaves it alone): the run builds a few objects per read and none of them
    in a cycle, but every ~70 000 of them trigger a full collection that walks the whole heap of the process -- torch, pandas and
    NumPy's modules included -- for 0.1-0.6 s each: 0.9 s of a 2.2 s run of 6 000 loci (scripts/exp_from_fast5.py, WSX_GC_DEBUG)."""
    import gc
    pools: list = []
    executor = None
    paused = len(loci) >= GC_PAUSE_FROM_LOCI and gc.isenabled() and not os.environ.get('WARPSTR_KEEP_GC')
    if paused:
        gc.disable()
    if threads and int(threads) > 1 and len(loci) > 1:
        from concurrent.futures import ThreadPoolExecutor
        executor = ThreadPoolExecutor(max_workers=min(int(threads), 64), initializer=spread_over_cpus)
        if len(loci) >= GC_PAUSE_FROM_LOCI:
            # (its threads started now, while nothing runs: the executor starts one per task handed in until it has them all, and a
            # thread start waits until the new thread has run -- 20-45 ms for sixteen of them once the first are busy setting loci up)
            import threading
            gate = threading.Barrier(executor._max_workers + 1)
            waiting = [executor.submit(gate.wait, 5.0) for _ in range(executor._max_workers)]
            try:
                gate.wait(5.0)
                for w in waiting:
                    w.result()
            except threading.BrokenBarrierError:
                pass   # (a machine that cannot start the threads in five seconds starts them when they are needed)
    try:
        return _main_wrapper_loci(loci, int(threads or 1), pools, executor, caller_config=caller_config, rescaler_config=rescaler_config,
                                  signal_loader=signal_loader, raw_reader=raw_reader, raw_reads=raw_reads, pore_model=pore_model,
                                  device=device, shard=shard, readers=readers, partition=partition, batch_reads=batch_reads, batch_samples=batch_samples,
                                  batch_raw_bytes=batch_raw_bytes, timings=timings, quiet=quiet, native=native, _engine=_engine)
    finally:  # the threads and the reader processes end with the call, however it ends
        if paused:
            gc.enable()
        from . import _readers
        while _readers._OPEN:
            _readers._OPEN.popitem()[1].close()
        if executor is not None:
            executor.shutdown(wait=True)
        for pool in pools:
            if pool is not None:
                pool.shutdown()


def _main_wrapper_loci(loci, threads, pools, executor, *, caller_config, rescaler_config, signal_loader, raw_reader, raw_reads, pore_model,
                       device, shard, readers, partition, batch_reads, batch_samples, batch_raw_bytes, timings, quiet, native, _engine):
    from . import dist as wdist
    from .pore_model import default_pore_model
    t_start = time.perf_counter()
    tm = timings if timings is not None else {}
    tm['_t0'], tm['_c0'] = t_start, time.process_time()
    for key in ('native_setup_s', 'overview_s', 'automata_s', 'similarity_s', 'handle_s', 'read_s', 'submit_s', 'collect_s', 'gather_s', 'store_s'):
        tm[key] = 0.0
    if partition not in ('auto', 'loci', 'reads'):
        raise ValueError("partition must be 'auto', 'loci' or 'reads'")
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
    pore_model = pore_model or default_pore_model()
    by_locus = collective and (partition == 'loci' or (partition == 'auto' and len(loci) >= LOCI_PER_RANK_FOR_LOCUS_PARTITION * world))
    tm['partition'] = 'loci' if by_locus else ('reads' if collective else 'none')
    tm['host_threads'] = threads if executor is not None else 1

    # ---- whose loci: all of them, or (partition by locus) this rank's ------------------------------------------------------
    if by_locus:
        own = partition_loci(loci, world)[rank]
    else:
        own = np.arange(len(loci))
    tm['loci_set_up'] = [int(i) for i in own]
    writes = rank == 0 or by_locus   # (by locus: a rank writes its own loci; by read: rank 0 writes everything)
    fast5_on_workers = signal_loader is None and raw_reader is read_raw_signal and raw_reads is None
    # (the reader processes start first: they come up -- half a second of imports -- while the loci are set up.  On the bench's
    # sandboxed box sixteen interpreters starting slow whatever runs beside them: the set-up 0.3 -> 0.9 s for 3 000 loci; started
    # beside the handle's creation instead they cost that 0.2 -> 1.5 s.)
    engine_probe = _engine or HipEngine
    zstd_on_device = (bool(getattr(engine_probe, 'DEVICE_ZSTD', False)) and hasattr(engine_probe, 'submit_vbz_parts')
                      and not os.environ.get('WARPSTR_NO_GPU_ZSTD') and not os.environ.get('WARPSTR_NO_GPU_VBZ') and not os.environ.get('WARPSTR_NO_READER_ARENAS'))
    pool = _reader_pool(threads, [loci[i] for i in own], readers, zstd_on_device) if fast5_on_workers else None
    _mark(tm, 'reader pool made')
    pools.append(pool)
    tm['reader_processes'] = pool._max_workers if pool is not None else 0

    # ---- per locus: overview, flanks, automata, state_similarity.csv --------------------------------------------------------
    error = None
    jobs: List[LocusJob] = []
    t0 = time.perf_counter()
    step = max(1, min(64, len(own) // (4 * max(tm['host_threads'], 1)) or 1))
    parts = [own[k:k + step] for k in range(0, len(own), step)]

    def setup(part):
        ptm: Dict[str, float] = {}
        t1 = time.perf_counter()
        chunk = [loci[i] for i in part]
        sts = _hostlib.NativeSetup.run_many([l.path for l in chunk], [l.sequence.upper() for l in chunk], pore_model,
                                            caller_config.min_state_similarity, writes) if native else None
        ptm['native_setup_s'] = time.perf_counter() - t1
        return [LocusJob(l, pore_model, ptm, caller_config, write=writes, native=native, setup=sts[q] if sts else None)
                for q, l in enumerate(chunk)], ptm

    # One rank, fast5 files, reader arenas and many loci: the set-up, the reading and the calling run as one pipeline
    # (_streamed_run) -- the readers start on the first loci's files while the later loci are still parsed and compiled, and the
    # handle takes their automata as they come.
    engine_cls = _engine or HipEngine
    probe = engine_cls if isinstance(engine_cls, type) else None
    streamed = None
    def print_warnings(part_jobs):
        if not quiet and writes:
            for job in part_jobs:
                for line in job.warnings:
                    print(line)
    can_stream = (not collective and len(own) >= STREAM_FROM_LOCI and probe is not None and hasattr(probe, 'add_automata')
                  and not os.environ.get('WARPSTR_NO_STREAMED_RUN'))
    if (can_stream and fast5_on_workers and all(hasattr(probe, a) for a in ('submit_raw_parts', 'ARENA_REGIONS', 'region_wait'))
            and os.path.isdir('/dev/shm') and not os.environ.get('WARPSTR_NO_READER_ARENAS')):
        workers = pool._max_workers if pool is not None else 1
        refused = _arena_room(workers, SHARED_BATCH_READS if pool is not None else SHARED_BATCH_READS // 4, batch_raw_bytes // 2)
        if refused is not None:
            tm['arenas_refused'] = refused
        else:
            # (with threads the parts are set up `threads` at a time and taken in order: executor.map's results -- started before the
            # reader processes are waited for: their helper takes 20 ms to fork them, the first loci are set up meanwhile)
            it = parts if executor is None else executor.map(setup, parts)
            pool = _started(pool, tm)
            _mark(tm, 'reader processes started')
            if pool is None:
                pool = _InlinePool()
                pools.append(pool)
            gpu_vbz = hasattr(probe, 'submit_vbz_parts') and not os.environ.get('WARPSTR_NO_GPU_VBZ')
            gpu_zstd = gpu_vbz and bool(getattr(probe, 'DEVICE_ZSTD', False)) and not os.environ.get('WARPSTR_NO_GPU_ZSTD')
            tm['reader_mode'] = ((('arenas, zstd and VBZ decoded on the GPU' if gpu_zstd else 'arenas, VBZ decoded on the GPU') if gpu_vbz else 'arenas')
                                 + (', filled in this process' if getattr(pool, 'inline', False) else '') + ', streamed with the set-up')

            streamed = _streamed_run(it, (lambda x: x) if executor is not None else setup, tm, pool, engine_cls,
                                     (caller_config, rescaler_config, local_gpu), batch_reads, batch_samples, batch_raw_bytes // 2, gpu_vbz,
                                     print_warnings, gpu_zstd)
    try:
        if streamed is not None:
            jobs = streamed[0]
        for part_jobs, ptm in (_thread_map(executor, setup, parts) if streamed is None else ()):
            jobs += part_jobs
            for key, v in ptm.items():
                tm[key] += v   # (CPU seconds, summed over the threads)
    except Exception as e:  # noqa: BLE001 -- agreed on below
        if not collective:
            raise
        error = e
    if streamed is None:
        tm['setup_wall_s'] = time.perf_counter() - t0
    if collective:
        wdist.agree_or_raise(error, world, coll_device, 'setting up the loci')
    if not quiet and writes and streamed is None:
        for job in jobs:
            for line in job.warnings:
                print(line)
    first = np.zeros(len(jobs) + 1, np.int64)
    np.cumsum([j.n for j in jobs], out=first[1:])
    n_total = int(first[-1])
    tm['n_loci'], tm['n_reads'] = len(jobs), n_total
    tm['native_overviews'] = sum(j.native is not None for j in jobs)

    records = np.zeros(0, dtype=_result_dtype())
    seq1 = seq2 = np.zeros(0, np.uint8)
    off1 = off2 = np.zeros(1, np.int64)
    if n_total == 0 and collective:
        wdist.agree_or_raise(None, world, coll_device, 'reading / calling the reads')   # (the other ranks are in this collective)
    if streamed is not None:
        _, _, records, seqs = streamed
        mine = np.arange(n_total)
        _mark(tm, 'all reads called')
    if n_total > 0 and streamed is None:
        # ---- the read list: (locus, row) -> automaton, cost ---------------------------------------------------------------
        counts = [j.n for j in jobs]
        locus_of = np.repeat(np.arange(len(jobs)), counts)
        row_of = np.concatenate([np.arange(c) for c in counts]) if jobs else np.zeros(0, np.int64)
        cat = lambda parts, dt: np.concatenate(parts) if parts else np.zeros(0, dt)
        reverse = cat([j.reverse for j in jobs], bool)
        lo, hi = cat([j.lo for j in jobs], np.int64), cat([j.hi for j in jobs], np.int64)
        aut = (2 * locus_of + reverse).astype(np.int32)
        span = (hi - lo + 1).clip(min=1)
        if collective and not by_locus:
            n_states = np.array([s.n_states for j in jobs for s in (j.temp_sta, j.rev_sta)])
            shards = wdist.shard_reads(span + wdist.READ_OVERHEAD_SAMPLES, world, np.array([wdist.slot_cost(s) for s in n_states])[aut])
        else:
            shards = [np.arange(n_total)] * max(rank + 1, 1)
        mine = shards[rank]
        whole = len(mine) == n_total
        all_names = None
        if raw_reads is not None:
            all_names = [nm for j in jobs for nm in j.names]

        # ---- rank-local: one handle, mixed-locus batches one behind the other ------------------------------------------
        error = None
        records = np.zeros(len(mine), dtype=_result_dtype())
        seqs = [[], []]
        try:
            import queue as _queue
            import threading
            queue = None
            engine_cls = _engine or HipEngine
            engine_ready, stop = threading.Event(), threading.Event()
            submitted: Dict[int, threading.Event] = {}   # arena batches: number -> "the calling thread has submitted it"
            handover: '_queue.Queue' = _queue.Queue(maxsize=1)

            def make_engine():
                nonlocal queue
                t0 = time.perf_counter()
                if len(mine):
                    queue = engine_cls([s for j in jobs for s in (j.temp_sta, j.rev_sta)],
                                       [j.flank_length for j in jobs for _ in range(2)], caller_config, rescaler_config, local_gpu)
                tm['handle_s'] += time.perf_counter() - t0
                engine_ready.set()
            # cut the rank's reads (in global order) into batches: at most batch_reads reads and batch_samples segment samples
            raw_budget = batch_raw_bytes // 2
            csum = np.cumsum(span[mine])
            cuts, a = [0], 0
            while a < len(mine):
                base = int(csum[a - 1]) if a else 0
                b = int(np.searchsorted(csum, base + batch_samples, side='right'))
                a = max(a + 1, min(a + batch_reads, b))
                cuts.append(a)
            pending = []  # (ticket, first, count)
            pool = _started(pool, tm)
            raw_len = np.full(len(mine), -1, np.int64)   # samples of a read's whole raw signal, once a reader process has said

            def item_of(k):   # what a reader needs to find read k of this rank: (annotated fast5, multi-read fall-back, read name)
                job, row = jobs[locus_of[mine[k]]], int(row_of[mine[k]])
                return (job.fast5_of(row), str(job.fast5_path[row]) if job.fast5_path is not None else None, job.names[row])

            def rows_of(x0, x1):
                """Reads [x0, x1) of this rank as the readers take them: one item per run of consecutive rows of a locus
                (LocusJob.rows_for_readers; _readers.expand makes item_of's items of it on the reader's CPU)."""
                g = mine[x0:x1]
                lj, rr = locus_of[g], row_of[g]
                cut = np.flatnonzero((lj[1:] != lj[:-1]) | (rr[1:] != rr[:-1] + 1)) + 1
                bounds = [0] + cut.tolist() + [x1 - x0]
                return [jobs[int(lj[a])].rows_for_readers(int(rr[a]), int(rr[a]) + z - a) for a, z in zip(bounds[:-1], bounds[1:])]

            def finish(ticket, b0, b1):
                t1 = time.perf_counter()
                rec, s1, p1, s2, p2 = queue.collect(ticket)
                tm['collect_s'] += time.perf_counter() - t1
                records[b0:b1] = rec
                seqs[0].append(s1)
                seqs[1].append(s2)

            def batches():
                """The rank's batches one after the other, read: (first, end, reads or [], staging slot or None, its read offsets).
                Run by the caller's thread, or -- with reader processes -- by a thread of its own, so that a batch is decoded while
                the previous one is submitted and an earlier one collected."""
                b = 0
                while b < len(cuts) - 1:
                    b0, b1 = cuts[b], cuts[b + 1]
                    t1 = time.perf_counter()
                    data, raw_bytes, shared_slot = [], 0, None
                    if raw_reads is not None:
                        sel_names = all_names[b0:b1] if whole else [all_names[g] for g in mine[b0:b1]]
                        data = [raw_reads[nm] for nm in sel_names]
                    elif pool is not None and b1 - b0 >= 64:
                        # the fast5 files of a batch on the worker processes
                        use_shared = hasattr(queue, 'stage_shared') and not tm.get('shared_staging_refused')
                        if use_shared and b1 - b0 > SHARED_BATCH_READS:   # (a shared batch is small: see SHARED_BATCH_BYTES)
                            cuts.insert(b + 1, b0 + SHARED_BATCH_READS)
                            b1 = b0 + SHARED_BATCH_READS
                        items = [item_of(k) for k in range(b0, b1)]
                        step = max(8, -(-len(items) // (CHUNKS_PER_READER * pool._max_workers)))   # (two chunks per worker: a round trip costs ~0.1 ms)
                        parts = [items[k:k + step] for k in range(0, len(items), step)]
                        if use_shared:
                            # two steps: lengths, then every read decoded to its place in a staging buffer both sides map
                            t2 = time.perf_counter()
                            todo = [k for k in range(len(items)) if raw_len[b0 + k] < 0]   # (a read is asked for its length once)
                            if todo:
                                tstep = max(8, -(-len(todo) // (2 * pool._max_workers)))
                                got = [n for part in pool.map(_probe_chunk, [[items[k] for k in todo[q:q + tstep]] for q in range(0, len(todo), tstep)])
                                       for n in part]
                                raw_len[b0 + np.asarray(todo)] = got
                            lens_b = raw_len[b0:b1].copy()
                            tm['probe_s'] = tm.get('probe_s', 0.0) + time.perf_counter() - t2
                            # (a shared batch is a fraction of the byte budget: the three staging buffers are then reused -- their pages
                            # are touched and page-locked once -- and the decoding of one batch runs beside the upload of the last)
                            keep = max(1, int(np.searchsorted(np.cumsum(lens_b) * 2, min(raw_budget, SHARED_BATCH_BYTES), side='right')))
                            if keep < len(items):
                                cuts.insert(b + 1, b0 + keep)
                                b1 = b0 + keep
                                items, lens_b = items[:keep], lens_b[:keep]
                                parts = [items[k:k + step] for k in range(0, len(items), step)]
                            shared_roff = np.zeros(len(items) + 1, np.int64)
                            np.cumsum(lens_b, out=shared_roff[1:])
                            try:
                                shared_slot = queue.stage_shared(int(shared_roff[-1]))
                            except OSError as e:   # no room under /dev/shm: the decoded reads come back through the pipes instead
                                print(f'warpstr_amd: no shared staging buffer ({e}); the reader processes return the reads through their pipes', file=sys.stderr)
                                tm['shared_staging_refused'] = str(e)
                                use_shared = False
                        if use_shared:
                            t2 = time.perf_counter()
                            # (the reads after this batch whose lengths are not known yet ride along: the next batch is then laid out
                            # without a round trip of its own)
                            ahead = [k for k in range(b1, min(len(mine), b1 + SHARED_BATCH_READS)) if raw_len[k] < 0]
                            n_parts = len(parts)
                            asked = [ahead[q::n_parts] for q in range(n_parts)]
                            answers = pool.map(_decode_chunk, [(shared_slot['path'], part, shared_roff[k:k + len(part)].tolist(), lens_b[k:k + len(part)].tolist(),
                                                                [item_of(x) for x in asked[q]])
                                                               for q, (k, part) in enumerate(zip(range(0, len(items), step), parts))])
                            for q, (busy, lens_ahead) in enumerate(answers):
                                tm['decode_worker_s'] = tm.get('decode_worker_s', 0.0) + float(busy)   # (summed over the reader processes)
                                if asked[q]:
                                    raw_len[np.asarray(asked[q])] = lens_ahead
                            tm['decode_s'] = tm.get('decode_s', 0.0) + time.perf_counter() - t2
                            tm['raw_bytes'] = tm.get('raw_bytes', 0) + int(shared_roff[-1]) * 2
                        else:
                            for part in pool.map(_read_chunk, parts):
                                data += part
                    elif fast5_on_workers and hasattr(queue, 'stage_local'):
                        # the fast5 files of a batch in this process, decoded straight to their places in the page-locked staging ring
                        # (a fresh array per read costs a page fault per 4 KiB -- more than HDF5, zstd and StreamVByte together)
                        if b1 - b0 > SHARED_BATCH_READS:
                            cuts.insert(b + 1, b0 + SHARED_BATCH_READS)
                            b1 = b0 + SHARED_BATCH_READS
                        items = [item_of(k) for k in range(b0, b1)]
                        lens_b = np.array(_probe_chunk(items), np.int64)
                        keep = max(1, int(np.searchsorted(np.cumsum(lens_b) * 2, min(raw_budget, SHARED_BATCH_BYTES), side='right')))
                        if keep < len(items):
                            cuts.insert(b + 1, b0 + keep)
                            b1 = b0 + keep
                            items, lens_b = items[:keep], lens_b[:keep]
                        shared_roff = np.zeros(len(items) + 1, np.int64)
                        np.cumsum(lens_b, out=shared_roff[1:])
                        shared_slot = queue.stage_local(int(shared_roff[-1]))
                        _decode_into(shared_slot['view'], items, shared_roff[:-1].tolist(), lens_b.tolist())
                        tm['raw_bytes'] = tm.get('raw_bytes', 0) + int(shared_roff[-1]) * 2
                    elif signal_loader is None:
                        for k in range(b0, b1):
                            g = mine[k]
                            data.append(jobs[locus_of[g]].raw_read(int(row_of[g]), raw_reader))
                    else:
                        for k in range(b0, b1):
                            g = mine[k]
                            data.append(np.asarray(signal_loader(jobs[locus_of[g]].fast5_of(int(row_of[g])), int(lo[g]), int(hi[g])), dtype=np.float64))
                    if signal_loader is None and shared_slot is None:
                        keep, acc = 0, 0  # long raw reads: as many as fit the byte budget, the rest open the next batch
                        while keep < len(data) and (keep == 0 or acc + data[keep].nbytes <= raw_budget):
                            acc += data[keep].nbytes
                            keep += 1
                        if keep < len(data):
                            cuts.insert(b + 1, b0 + keep)
                            b1 = b0 + keep
                            del data[keep:]
                    tm['read_s'] += time.perf_counter() - t1
                    yield b0, b1, data, shared_slot, (shared_roff if shared_slot is not None else None)
                    b += 1

            def arena_batches():
                """The same with reader processes that decode into arenas of their own (_readers.decode_arena): nothing is asked of
                a read before it is decoded, so the chunks of the next batch are handed out while this batch's slowest chunk is
                still running -- the readers never wait for each other, only a region for the upload of the batch that used it
                three batches ago."""
                import collections
                regions = engine_cls.ARENA_REGIONS
                inflight = collections.deque()
                b, k, ci = 0, 0, 1
                # A batch is also cut at its byte budget (batch_raw_bytes / 2, as batches() cuts): nothing is known about a read
                # before it is decoded except that it reaches its segment's end -- 2 (r_end_raw + 1) bytes at least --, so a read
                # counts as that or as the mean of the reads decoded so far, whichever is more (ultra-long reads: a batch of 2 048
                # of them would ask for gigabytes of arena, device buffer and page-locked memory per region)
                least = 2 * (hi[mine] + 1).clip(min=1)
                seen = [0, 0]   # reads decoded so far, their bytes
                while b < len(mine) or inflight:
                    while b < len(mine) and len(inflight) < regions - 1:
                        while cuts[ci] <= b:
                            ci += 1
                        b1 = min(cuts[ci], b + (SHARED_BATCH_READS // 4 if getattr(pool, 'inline', False) else SHARED_BATCH_READS))
                        mean = seen[1] // seen[0] if seen[0] else 0
                        if not seen[0]:   # (nothing decoded yet: a small batch tells what this run's reads weigh)
                            b1 = min(b1, b + ARENA_PROBE_READS)
                        fit = int(np.searchsorted(np.cumsum(np.maximum(least[b:b1], mean)), raw_budget, side='right'))
                        b1 = b + max(1, min(b1 - b, fit))
                        t1 = time.perf_counter()
                        region = k % regions
                        if k >= regions:   # (a region's first use waits for nothing -- and the handle may still be in the making)
                            # the batch that used the region before must have been SUBMITTED by the calling thread (its copies
                            # enqueued, the region's event recorded: taking it from the hand-over queue is not enough) ...
                            while not submitted[k - regions].wait(0.2):
                                if stop.is_set():
                                    return
                            if stop.is_set():
                                return
                            queue.region_wait(region)   # ... and uploaded
                            del submitted[k - regions]
                        submitted[k] = threading.Event()
                        step = max(8, -(-(b1 - b) // (CHUNKS_PER_READER * pool._max_workers)))
                        futures = []
                        for q in range(b, b1, step):
                            items = rows_of(q, min(q + step, b1))
                            futures.append(pool.submit(_pack_arena if gpu_vbz else _decode_arena, (region, k, items, gpu_zstd) if gpu_vbz else (region, k, items)))
                        inflight.append((b, b1, region, futures, k))
                        tm['read_s'] += time.perf_counter() - t1
                        b, k = b1, k + 1
                    b0, b1, region, futures, kb = inflight.popleft()
                    t1 = time.perf_counter()
                    parts = []
                    try:
                        answers = [f.result() for f in futures]
                    except RuntimeError as e:
                        if 'no room for a reader arena' not in str(e):
                            raise
                        # /dev/shm filled up under the run (another job's files, reads far longer than ARENA_ROOM_PER_READ): this
                        # batch's reads come back through the pipes instead and go up from the page-locked ring (said once)
                        if not tm.get('arena_fallbacks'):
                            print(f'warpstr_amd: {str(e).strip().splitlines()[-1]}; such batches are read without arenas', file=sys.stderr)
                        tm['arena_fallbacks'] = tm.get('arena_fallbacks', 0) + 1
                        for f in futures:   # (the chunks that did fit have written their part of the region: nothing to undo)
                            f.exception()
                        data = []
                        if getattr(pool, 'inline', False):
                            data = _read_chunk([item_of(x) for x in range(b0, b1)])
                        else:
                            items = [item_of(x) for x in range(b0, b1)]
                            step = max(8, -(-len(items) // (CHUNKS_PER_READER * pool._max_workers)))
                            for part in pool.map(_read_chunk, [items[q:q + step] for q in range(0, len(items), step)]):
                                data += part
                        seen[0] += b1 - b0
                        seen[1] += int(sum(d.nbytes for d in data))
                        tm['raw_bytes'] = tm.get('raw_bytes', 0) + int(sum(d.nbytes for d in data))
                        submitted[kb].set()   # (no region of the arenas is in use by this batch)
                        tm['read_s'] += time.perf_counter() - t1
                        yield b0, b1, data, None, None
                        continue
                    seen[0] += b1 - b0
                    for got in answers:
                        if gpu_vbz:
                            path, cap, base, used, lens_p, table, busy = got
                            cap //= 2   # (the arena's size in samples, as the engine counts it)
                            parts.append((path, 2 * cap, base, used, lens_p, table))
                            tm['uploaded_bytes'] = tm.get('uploaded_bytes', 0) + int(used)
                        else:
                            path, cap, base, lens_p, busy = got
                            parts.append((path, cap, base, lens_p))
                            tm['uploaded_bytes'] = tm.get('uploaded_bytes', 0) + 2 * int(sum(lens_p))
                        page_lock([(path, cap)])   # (here, on the reader thread, while the other chunks are still decoding)
                        tm['decode_worker_s'] = tm.get('decode_worker_s', 0.0) + float(busy)
                        tm['raw_bytes'] = tm.get('raw_bytes', 0) + 2 * int(sum(lens_p))
                        seen[1] += 2 * int(sum(lens_p))
                    tm['decode_s'] = tm.get('decode_s', 0.0) + time.perf_counter() - t1
                    tm['read_s'] += time.perf_counter() - t1
                    yield b0, b1, parts, ('vbz' if gpu_vbz else 'arena', region, kb), None

            def page_lock(parts):
                """Map and page-lock the arenas the parts lie in, if the engine is there to do it (else the submitting thread does it
                when it uploads from them); True once done."""
                if not engine_ready.is_set():
                    return False
                ready = getattr(queue, 'arena_ready', None)
                if ready is not None:
                    for part in parts:
                        ready(part[0], part[1] // 2 if len(part) == 6 else part[1])   # (a part of packed blocks counts bytes)
                return True

            def submit(b0, b1, data, shared_slot, shared_roff):
                t1 = time.perf_counter()
                sel = mine[b0:b1]
                if isinstance(shared_slot, tuple) and shared_slot[0] == 'vbz':
                    ticket = queue.submit_vbz_parts(shared_slot[1], data, lo[sel], hi[sel], aut[sel])
                elif isinstance(shared_slot, tuple) and shared_slot[0] == 'arena':
                    ticket = queue.submit_raw_parts(shared_slot[1], data, lo[sel], hi[sel], aut[sel])
                elif shared_slot is not None:
                    ticket = queue.submit_raw_shared(shared_slot, shared_roff, lo[sel], hi[sel], aut[sel])
                elif signal_loader is None:
                    ticket = queue.submit_raw(data, lo[sel], hi[sel], aut[sel])
                else:
                    ticket = queue.submit_signals(data, aut[sel])
                tm['submit_s'] += time.perf_counter() - t1
                if isinstance(shared_slot, tuple) and len(shared_slot) == 3:
                    submitted[shared_slot[2]].set()   # (the reader thread may hand the batch's arena region out again)
                pending.append((ticket, b0, b1))
                if len(pending) > 2:  # at most three batches' buffers in HBM / in flight
                    finish(*pending.pop(0))

            def produce(source):
                try:
                    for item in source():
                        locked = not (isinstance(item[3], tuple) and item[3][0] in ('arena', 'vbz'))
                        while not stop.is_set():
                            # (a batch decoded before the handle existed: its arenas are page-locked while it waits its turn)
                            locked = locked or page_lock(item[2])
                            try:
                                handover.put(item, timeout=0.2 if locked else 0.01)
                                break
                            except _queue.Full:
                                pass
                        if stop.is_set():
                            return
                    item = None
                except BaseException as e:  # noqa: BLE001 -- raised by the consumer below
                    item = e
                while not stop.is_set():
                    try:
                        handover.put(item, timeout=0.2)
                        return
                    except _queue.Full:
                        pass

            # With reader processes a reader thread runs one batch ahead of the calling thread: shared staging has three slots, and a
            # slot is taken again only after the batch that used it was submitted (the hand-over queue holds one batch: taking the
            # slot of batch n + 3 follows putting batch n + 2, which follows the consumer's get of batch n + 1, i.e. its submit of
            # batch n).  With arenas nothing of the handle is needed to decode the first batches: the reader thread starts BEFORE the
            # handle is created, and the readers decode while 12 000 automata are placed and packed.
            probe = engine_cls if isinstance(engine_cls, type) else None   # (what the engine can do, asked of its class)
            arenas = (probe is not None and hasattr(probe, 'submit_raw_parts') and hasattr(probe, 'ARENA_REGIONS')
                      and os.path.isdir('/dev/shm') and not os.environ.get('WARPSTR_NO_READER_ARENAS') and len(mine) > 0)
            if arenas and (pool is not None or fast5_on_workers):
                # room for the arenas?  Three regions of a batch each -- ARENA_ROOM_PER_READ bytes a read, 16 MB a reader process at
                # least -- and half as much again; a container's default /dev/shm of 64 MB has not, and the run then takes the
                # staging ring (which falls back to the pipes by itself) or, in one process, the page-locked ring
                workers = pool._max_workers if pool is not None else 1
                reads = min(len(mine), SHARED_BATCH_READS if pool is not None else SHARED_BATCH_READS // 4)
                refused = _arena_room(workers, reads, raw_budget)
                if refused is not None:
                    tm['arenas_refused'] = refused
                    arenas = False
            if pool is None and arenas and fast5_on_workers:
                # no reader processes (one thread, few loci, or they could not be started): the same arenas, filled by the reader
                # thread itself -- a read is opened once (no pass for the lengths first) and decoded beside the calling thread
                pool = _InlinePool()
                pools.append(pool)
            arenas = arenas and pool is not None
            # (and if the engine decodes VBZ itself -- wsx_vbz_decode -- the readers stop at the zstd frame's content: StreamVByte,
            # zig-zag and the running sum are a quarter of a reader's time per read and 0.6 of the bytes to upload)
            gpu_vbz = arenas and hasattr(probe, 'submit_vbz_parts') and not os.environ.get('WARPSTR_NO_GPU_VBZ')
            # (... and if it decodes zstd as well -- wsx_zstd_decode -- a reader's part of a read is libhdf5 alone)
            gpu_zstd = gpu_vbz and bool(getattr(probe, 'DEVICE_ZSTD', False)) and not os.environ.get('WARPSTR_NO_GPU_ZSTD')
            reader = None
            try:
                if arenas:
                    tm['reader_mode'] = ((('arenas, zstd and VBZ decoded on the GPU' if gpu_zstd else 'arenas, VBZ decoded on the GPU') if gpu_vbz else 'arenas')
                                         + (', filled in this process' if getattr(pool, 'inline', False) else ''))
                    reader = threading.Thread(target=produce, args=(arena_batches,), name='warpstr-reader', daemon=True)
                    reader.start()
                make_engine()
                if reader is None and pool is not None and hasattr(queue, 'stage_shared'):
                    tm['reader_mode'] = 'shared staging'
                    reader = threading.Thread(target=produce, args=(batches,), name='warpstr-reader', daemon=True)
                    reader.start()
                if reader is not None:
                    while True:
                        item = handover.get()
                        if item is None:
                            break
                        if isinstance(item, BaseException):
                            raise item
                        submit(*item)
                else:
                    for item in batches():
                        submit(*item)
            finally:
                stop.set()
                engine_ready.set()
                if reader is not None:
                    reader.join()
            while pending:
                finish(*pending.pop(0))
            if queue is not None:
                tm.update(queue.info())
                queue.close()
        except Exception as e:  # noqa: BLE001 -- agreed on below: no rank may wait in a collective for one that failed
            error = e
        if collective or error is not None:
            wdist.agree_or_raise(error, world if collective else 1, coll_device, 'reading / calling the reads')

    if n_total > 0:
        # ---- the complete table of this rank's loci ---------------------------------------------------------------------------
        t0 = time.perf_counter()
        ok = records['status'] == 0
        l1 = np.where(ok, records['len1'], 0).astype(np.int64)
        l2 = np.where(ok, records['len2'], 0).astype(np.int64)
        off1, off2 = np.zeros(len(mine) + 1, np.int64), np.zeros(len(mine) + 1, np.int64)
        np.cumsum(l1, out=off1[1:])
        np.cumsum(l2, out=off2[1:])
        seq1 = np.concatenate(seqs[0]) if seqs[0] else np.zeros(0, np.uint8)
        seq2 = np.concatenate(seqs[1]) if seqs[1] else np.zeros(0, np.uint8)
        if collective and not by_locus:
            local = CallerResults([], records, off1[:-1], seq1, seq2, 'nan', offsets2=off2[:-1])
            records, seq1, o1, seq2, o2 = wdist.gather_called(local, mine, shards, n_total, world, coll_device)
            off1, off2 = np.append(o1, len(seq1)), np.append(o2, len(seq2))
        tm['gather_s'] += time.perf_counter() - t0

    # ---- per locus: the outputs of main_wrapper -------------------------------------------------------------------------------
    t0 = time.perf_counter()
    # the first read a caller failed on ends the run where upstream's loop would have ended: the loci before it are written
    bad = np.flatnonzero(records['status'] != 0)
    n_good = int(np.searchsorted(first, bad[0], side='right') - 1) if len(bad) else len(jobs)
    first_bad = int(own[n_good]) if n_good < len(jobs) else len(loci)   # (index in `loci`)
    if by_locus:
        first_bad = int(wdist.gather_counts(first_bad, world, coll_device).min())
        n_good = int(np.searchsorted(own, first_bad))

    def store(li, overview_done=False):
        job = jobs[li]
        a, b = int(first[li]), int(first[li + 1])
        s1, s2 = seq1[int(off1[a]):int(off1[b])], seq2[int(off2[a]):int(off2[b])]
        return _store_job(job, records[a:b], s1, off1[a:b] - off1[a], s2, off2[a:b] - off2[a], writes, quiet, overview_done)

    # the run's per-read columns, contiguous: a chunk of native loci writes its overview.csv / FASTA files in ONE library call
    cols = None
    if n_good and _hostlib.lib() is not None:
        cols = (np.ascontiguousarray(records['len1'], np.int32), np.ascontiguousarray(records['len2'], np.int32),
                np.ascontiguousarray(records['cost1'], np.float64), np.ascontiguousarray(records['cost2'], np.float64),
                np.ascontiguousarray(seq2, np.uint8), np.ascontiguousarray(off2[:-1] if len(off2) > len(records) else off2, np.int64))

    def store_chunk(ids):
        _hostlib.store_many([jobs[li].native for li in ids], [jobs[li].locus.path for li in ids], [first[li] for li in ids], *cols, writes)
        return [store(li, True) for li in ids]

    entries: list = [None] * len(loci)
    error = None
    try:
        native_ids = [li for li in range(n_good) if jobs[li].native is not None] if cols is not None else []
        step = max(1, min(64, len(native_ids) // (4 * max(tm['host_threads'], 1)) or 1))
        done = {}
        for ids, part in zip([native_ids[k:k + step] for k in range(0, len(native_ids), step)],
                             _thread_map(executor, store_chunk, [native_ids[k:k + step] for k in range(0, len(native_ids), step)])):
            done.update(zip(ids, part))
        for li in range(n_good):  # (tables that went through pandas: here, one after the other)
            entry, messages = done[li] if li in done else store(li)
            entries[int(own[li])] = entry
            if not quiet and writes:
                for line in messages:
                    print(line)
        if n_good < len(jobs) and int(own[n_good]) == first_bad:
            store(n_good)   # raises upstream's error for the locus's first failed read
    except Exception as e:  # noqa: BLE001
        if not collective:
            raise
        error = e
    tm['store_s'] += time.perf_counter() - t0
    _mark(tm, 'outputs written')
    if collective:
        wdist.agree_or_raise(error, world, coll_device, 'writing the outputs')
        if first_bad < len(loci):  # (another rank's locus: it has raised ReadCallError, agree_or_raise named it here)
            raise ReadCallError(f'a read of locus {getattr(loci[first_bad], "path", first_bad)} could not be called')
        tdist.barrier()  # the files are complete when any rank returns
    for i in range(len(loci)):
        if entries[i] is None:   # another rank's locus: its files say everything
            units_gt1 = sum(ch == '(' for ch in loci[i].sequence) > 0
            entries[i] = ('disk', loci[i].path, units_gt1)
    tm['total_s'] = time.perf_counter() - t_start
    tm.pop('_t0', None), tm.pop('_c0', None)
    return LociTables(entries)


def _result_dtype():
    from . import _lib
    return _lib.RESULT_DTYPE
