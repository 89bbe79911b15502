"""What a reader process of main_wrapper_loci runs (warpstr_amd/_hostworker.py; loci.py: _WorkerPool): the fast5 files of a batch.
This module imports NumPy and the fast5 reader only -- a worker is up in the time the parent needs to parse its overviews.

Items are (annotated single-read fast5 path, multi-read fall-back path or None, read name) triples, resolved the way
LocusJob.raw_read / wrapper.get_raw_workload resolve them (src/caller/wrapper.py:44-54; prepare_caller_only.py keeps the reads
of a caller-only input in their multi-read files)."""
import os
from typing import Dict

import numpy as np

_OPEN: Dict[str, object] = {}    # per process: fast5 path -> open Fast5File (a batch reads many reads of few files)
_MAPS: Dict[str, tuple] = {}     # per process: staging path -> (mmap, int16 view)
MAX_OPEN = 64


def fast5_file(path: str):
    from .fast5 import Fast5File
    f = _OPEN.get(path)
    if f is None:
        if len(_OPEN) >= MAX_OPEN:
            _OPEN.pop(next(iter(_OPEN))).close()
        f = _OPEN[path] = Fast5File(path)
    return f


def resolve(item):
    path, fallback, name = item
    if not os.path.exists(path) and fallback is not None:
        return fallback, name     # caller-only input: the read is still in its multi-read file
    return path, None


def read_chunk(items):
    """The raw reads (int16) of the items, returned through the pipe."""
    out = []
    for item in items:
        path, read_id = resolve(item)
        out.append(np.ascontiguousarray(fast5_file(path).raw_signal(read_id), dtype=np.int16))
    return out


# The same in two steps, without the decoded samples going through a pipe: the workers first say how long their reads are
# (metadata), the parent lays the batch out in a staging buffer both sides map (caller.SharedStaging), the workers then decode
# each read straight to its place -- and the GPU upload starts from that buffer.
def probe_chunk(items):
    """Samples of each read; nothing is decoded."""
    out = []
    for item in items:
        path, read_id = resolve(item)
        out.append(fast5_file(path).signal_length(read_id))
    return out


def decode_into(view: np.ndarray, items, offsets, lengths) -> float:
    """Decode each read into view[offset : offset + length] (an int16 staging buffer); returns the seconds it took."""
    import time
    t0 = time.perf_counter()
    for item, off, n in zip(items, offsets, lengths):
        path, read_id = resolve(item)
        fast5_file(path).raw_signal_into(read_id, view[off:off + n])
    return time.perf_counter() - t0


def decode_chunk(args):
    """decode_into() for the staging FILE both processes map (caller.SharedStaging), by its path.  A fifth element: items of the
    NEXT batch whose lengths the parent does not know yet -- answered in the same round trip (the parent then lays that batch out
    without a round of its own).  Returns (seconds spent decoding, [their lengths])."""
    import mmap
    staging, items, offsets, lengths = args[:4]
    got = _MAPS.get(staging)
    if got is None or len(got[1]) < max((o + n for o, n in zip(offsets, lengths)), default=0):
        for key in [k for k in _MAPS if not os.path.exists(k)]:
            _MAPS.pop(key)
        with open(staging, 'r+b') as fh:
            size = os.fstat(fh.fileno()).st_size
            mm = mmap.mmap(fh.fileno(), size)   # (not pre-faulted: a worker writes a sixteenth of it)
        got = _MAPS[staging] = (mm, np.frombuffer(mm, dtype=np.int16))
    busy = decode_into(got[1], items, offsets, lengths)
    return busy, (probe_chunk(args[4]) if len(args) > 4 and args[4] else [])


def spread_over_cpus(k: int):
    """Move this process to the k-th CPU of its affinity mask and release it again (loci.spread_over_cpus, without the package)."""
    if not hasattr(os, 'sched_setaffinity'):
        return
    try:
        allowed = sorted(os.sched_getaffinity(0))
        if len(allowed) >= 2:
            k += int(os.environ.get('LOCAL_RANK', '0') or 0) * 16
            os.sched_setaffinity(0, {allowed[k % len(allowed)]})
            os.sched_setaffinity(0, allowed)
    except OSError:
        pass
