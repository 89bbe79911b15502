"""What a reader process of main_wrapper_loci runs (warpstr_amd/_hostworker.py; loci.py: _WorkerPool): the fast5 files of a batch.
This module imports the NumPy-free core of the fast5 reader (_h5core) and nothing else until a function that hands arrays around is
called: the arena path (decode_arena) never is one, so sixteen readers are up in ~80 ms instead of ~0.7 s.

Items are (annotated single-read fast5 path, multi-read fall-back path or None, read name) triples, resolved the way
LocusJob.raw_read / wrapper.get_raw_workload resolve them (src/caller/wrapper.py:44-54; prepare_caller_only.py keeps the reads
of a caller-only input in their multi-read files)."""
import os
from typing import Dict

_OPEN: Dict[str, object] = {}    # per process: fast5 path -> open Fast5File (a batch reads many reads of few files)
_MAPS: Dict[str, tuple] = {}     # per process: staging path -> (mmap, int16 view)
MAX_OPEN = 64


def fast5_file(path: str, arrays: bool = True):
    """The open file of a path; arrays=False: its NumPy-free core is enough (lengths, decode_to an address)."""
    f = _OPEN.get(path)
    if f is not None and arrays and not hasattr(f, 'raw_signal'):
        _OPEN.pop(path).close()
        f = None
    if f is None:
        if len(_OPEN) >= MAX_OPEN:
            _OPEN.pop(next(iter(_OPEN))).close()
        if arrays:
            from .fast5 import Fast5File as cls
        else:
            from ._h5core import Fast5Core as cls
        f = _OPEN[path] = cls(path)
    return f


ROWS = '\0rows'   # first element of an item that stands for several reads of one locus (loci.LocusJob.rows_for_readers)


def expand(items):
    """The per-read items (annotated fast5 path, multi-read fall-back or None, read name) of what the parent sent: such items as
    they are, a locus's rows -- (ROWS, {run id: directory of its annotated files}, [run id] or None, [name], [fall-back] or None) --
    put together here, on the reader's CPU (upstream's path of a read: src/caller/wrapper.py:49-50)."""
    if not any(len(it) == 5 and it[0] == ROWS for it in items):
        return items
    out = []
    join = os.path.join
    for it in items:
        if len(it) == 5 and it[0] == ROWS:
            _, dirs, runs, names, falls = it
            only = dirs['0'] if runs is None else None
            for k, name in enumerate(names):
                out.append((join(only if runs is None else dirs[runs[k]], name + '.fast5'), falls[k] if falls is not None else None, name))
        else:
            out.append(it)
    return out


def resolve(item):
    path, fallback, name = item
    if not os.path.exists(path) and fallback is not None:
        return fallback, name     # caller-only input: the read is still in its multi-read file
    return path, None


def read_chunk(items):
    """The raw reads (int16) of the items, returned through the pipe."""
    import numpy as np
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
        out.append(fast5_file(path, arrays=False).signal_length(read_id))
    return out


def decode_into(view, items, offsets, lengths) -> float:
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

    import numpy as np
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


# ---- arenas: one phase, no barrier -------------------------------------------------------------------------------------------
# A reader decodes a chunk of reads back to back into a memory-backed file of ITS OWN (one per region: the parent hands out the
# regions in turn and lets a region be written again only after what it held has been uploaded) and answers with the lengths.
# Nothing has to be known about a read before it is decoded, so the parent can hand out the next batch's chunks while this one's
# slowest chunk is still running: the readers never wait for each other.  The parent maps the same files and page-locks them.
_CURSOR: Dict[int, list] = {}         # region -> [generation, samples written in it] (decode_arena)
_BYTE_CURSOR: Dict[int, list] = {}    # region -> [generation, bytes written in it] (pack_arena)
_ARENAS: Dict[int, list] = {}     # region -> [path, mmap, its first byte as a ctypes object (keeps the address valid), samples]
ARENA_DIR = '/dev/shm'


def _arena(region: int, samples: int):
    """The region's file mapped with room for `samples` samples (grown by half as much again; at least 8 M samples)."""
    import atexit
    import ctypes
    import mmap
    import tempfile
    cur = _ARENAS.get(region)
    if cur is not None and cur[3] >= samples:
        return cur
    cap = max(samples + samples // 2, 8 << 20)
    if cur is None:
        fd, path = tempfile.mkstemp(prefix=f'warpstr_arena_{os.getpid()}_{region}_', dir=ARENA_DIR)
        if not _ARENAS:
            atexit.register(_drop_arenas)
    else:
        path = cur[0]
        cur[2] = None            # (unmap before the file grows: the export first, then the mapping)
        cur[1].close()
        cur[1], cur[3] = None, 0
        fd = os.open(path, os.O_RDWR)
    try:
        st = os.statvfs(ARENA_DIR)
        if st.f_bavail * st.f_frsize < cap * 2 + (64 << 20):
            raise OSError(f'{ARENA_DIR} has no room for a reader arena of {cap * 2} bytes (WARPSTR_NO_READER_ARENAS=1 reads without arenas)')
        os.ftruncate(fd, cap * 2)
        mm = mmap.mmap(fd, cap * 2)
    finally:
        os.close(fd)
    _ARENAS[region] = [path, mm, ctypes.c_char.from_buffer(mm), cap]
    return _ARENAS[region]


def prepare_arenas(regions, samples: int = 8 << 20):
    """Create the arenas of `regions` ahead of the first task and touch their pages (a reader process does this while the parent
    still sets its loci up: the first write into a fresh page of a memory-backed file costs a fault, ~50 of them per read).  Best
    effort: no room, no arenas."""
    try:
        for region in regions:
            rec = _arena(region, samples)
            view = memoryview(rec[1])
            for at in range(0, len(view), 4096):
                view[at] = 0
            view.release()
    except (OSError, ValueError):
        pass


def _arena_address(region: int, samples: int) -> int:
    import ctypes
    return ctypes.addressof(_arena(region, samples)[2])


def _drop_arenas(owner=None):
    """Unlink and forget the arenas (of the regions whose key starts with `owner`: a run that filled arenas in its own process
    drops its own -- loci._InlinePool -- and leaves those of a run on another thread alone)."""
    for key in [k for k in _ARENAS if owner is None or str(k).startswith(owner)]:
        rec = _ARENAS.pop(key)
        try:
            os.unlink(rec[0])
        except OSError:
            pass
        _CURSOR.pop(key, None)
        _BYTE_CURSOR.pop(key, None)


def decode_arena(args):
    """(region, generation, items) -> (arena path, its size in samples, where this chunk starts, [length of every read], seconds):
    the reads decoded back to back behind whatever this process has already written into the region in this generation (a batch may
    give a reader several chunks); a new generation starts at the region's beginning.  No NumPy on this path (_h5core.decode_to
    writes at an address) unless a file holds chunks only fast5.py's decoders read."""
    import time

    from ._h5core import NeedsNumpy
    region, generation, items = args
    t0 = time.perf_counter()
    items = expand(items)
    cur = _CURSOR.setdefault(region, [generation, 0])
    if cur[0] != generation:
        cur[0], cur[1] = generation, 0
    base = at = cur[1]
    lens = []

    def place(n):   # the destination of the next read, once the reader knows its length
        addr = _arena_address(region, at + n) + 2 * at   # (growing keeps what is there: the file is the memory)
        lens.append(int(n))
        return addr
    for item in items:
        path, read_id = resolve(item)
        done = len(lens)
        try:
            fast5_file(path, arrays=False).decode_to(read_id, place)
        except NeedsNumpy:
            import ctypes
            del lens[done:]
            raw = fast5_file(path).raw_signal(read_id)
            ctypes.memmove(place(len(raw)), raw.ctypes.data, 2 * len(raw))
        at += lens[-1]
    cur[1] = at
    arena = _arena(region, max(at, 1))
    return arena[0], arena[3], base, lens, time.perf_counter() - t0


_NATIVE = False   # (library, or None) once asked


def _native_reader():
    """warpstr_amd/_host_loci.so's reader loop (csrc/host_reader.cpp: wsh_reader_pack), bound to the libhdf5 / libzstd this process
    uses -- or None: the library is not built, lacks the export (a stale build), HDF5 is older than 1.10.5, or
    WARPSTR_NO_HOST_NATIVE / WARPSTR_NO_NATIVE_READER say so.  The ctypes reader below then does the same work."""
    global _NATIVE
    if _NATIVE is False:
        _NATIVE = None
        import ctypes as C
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_host_loci.so')
        if os.path.exists(path) and not os.environ.get('WARPSTR_NO_HOST_NATIVE') and not os.environ.get('WARPSTR_NO_NATIVE_READER'):
            try:
                from ._h5core import lib_paths, libs, thread_context, vbz_native
                libs()                      # (libhdf5 initialised, its error stack silenced, as the ctypes reader has it)
                lib = C.CDLL(path)
                fn = lib.wsh_reader_pack
                fn.restype = C.c_int64
                fn.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_void_p,
                               C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int32]
                lib.wsh_reader_init.argtypes = [C.c_char_p, C.c_char_p]
                h5, zs = lib_paths()
                if lib.wsh_reader_init(os.fsencode(h5), os.fsencode(zs)) == 0:
                    vbz_native()            # (binds the per-thread zstd context wsh_vbz_unpack uses)
                    _NATIVE = (lib, thread_context)
            except (OSError, AttributeError, RuntimeError):
                _NATIVE = None
    return _NATIVE


def _pack_native(region: int, at: int, items, device_zstd: bool = False):
    """pack_arena's loop in one library call per chunk (more when the arena has to grow): (bytes written up to, [samples of every
    read], the block table as bytes) or None when a read needs the ctypes reader (another layout or filter, any error: that path
    then raises what it finds) -- nothing of the chunk counts as written in that case."""
    native = _native_reader()
    if native is None:
        return None
    import ctypes as C
    lib, thread_context = native
    thread_context()
    n = len(items)
    enc = os.fsencode
    paths = (C.c_char_p * n)(*[enc(it[0]) for it in items])
    falls = (C.c_char_p * n)(*[enc(it[1]) if it[1] is not None else None for it in items])
    names = (C.c_char_p * n)(*[it[2].encode('utf-8') if it[2] is not None else None for it in items])
    lens, status = (C.c_int64 * n)(), (C.c_int32 * n)()
    table_cap = 2 * n + 16
    table = (C.c_int64 * (7 * table_cap))()
    pos, n_blocks, need = C.c_int64(at), C.c_int64(0), C.c_int64(0)
    arena = _arena(region, max((at + 1) // 2, 1))
    first = 0
    while first < n:
        got = lib.wsh_reader_pack(first, n, paths, falls, names, C.addressof(arena[2]), 2 * arena[3], C.byref(pos), lens, status, table,
                                  table_cap, C.byref(n_blocks), C.byref(need), 1 if device_zstd else 0)
        if got < 0 or (got < n and status[got] != 0):
            return None
        if got < n:   # read `got` did not fit: more room, then on from there
            if need.value == 0:
                bigger = (C.c_int64 * (14 * table_cap))()
                C.memmove(bigger, table, 56 * n_blocks.value)
                table, table_cap = bigger, 2 * table_cap
            else:
                arena = _arena(region, (need.value + 1) // 2 + 8)   # (growing keeps what is there: the file is the memory)
        first = got
    return pos.value, list(lens), C.string_at(table, 56 * n_blocks.value)


def pack_arena(args):
    """decode_arena() for a parent that decodes on the GPU (wsx_vbz_decode): every read's blocks -- StreamVByte blocks as they
    leave zstd, or plain samples -- back to back (16-byte aligned) in the region's arena.  (region, generation, items) ->
    (arena path, its size in bytes, first byte of this chunk, bytes used from there, [samples of every read],
    seven int64 per block (read of the chunk, kind, first byte in the arena, bytes, samples wanted, values coded, content bytes) of all
    blocks as bytes, seconds).  A fourth element of the argument, if true: the parent decodes zstd on the device too
    (wsx_zstd_decode) -- a chunk whose frame that decoder takes is left as the frame lies in the file (kinds 3 / 4, content bytes =
    what the frame declares), and the reader's part of a read is libhdf5 alone."""
    import array
    import time

    from ._h5core import PLAIN, NeedsNumpy
    region, generation, items = args[:3]
    device_zstd = bool(args[3]) if len(args) > 3 else False
    t0 = time.perf_counter()
    items = expand(items)
    cur = _BYTE_CURSOR.setdefault(region, [generation, 0])
    if cur[0] != generation:
        cur[0], cur[1] = generation, 0
    base = at = cur[1]
    done = _pack_native(region, at, items, device_zstd)
    if done is not None:
        at, lens, table_bytes = done
        cur[1] = at
        arena = _arena(region, max((at + 1) // 2, 1))
        return arena[0], 2 * arena[3], base, at - base, lens, table_bytes, time.perf_counter() - t0
    lens, table, offs = [], array.array('q'), []

    def place(nbytes):   # where the next block goes
        nonlocal at
        at = (at + 15) & ~15
        addr = _arena_address(region, (at + nbytes + 1) // 2 + 8) + at
        offs.append(at)
        at += nbytes
        return addr
    for r, item in enumerate(items):
        path, read_id = resolve(item)
        mark, n_offs = at, len(offs)
        try:
            n, blocks = fast5_file(path, arrays=False).blocks_to(read_id, place, device_zstd)
        except NeedsNumpy:
            import ctypes
            at = mark
            del offs[n_offs:]
            raw = fast5_file(path).raw_signal(read_id)
            ctypes.memmove(place(2 * len(raw)), raw.ctypes.data, 2 * len(raw))
            n, blocks = len(raw), [(PLAIN, 2 * len(raw), len(raw), len(raw), 0)]
        lens.append(int(n))
        for (kind, nbytes, ns, nv, content), off in zip(blocks, offs[n_offs:]):
            table.extend((r, kind, off, nbytes, ns, nv, content))
    cur[1] = at
    arena = _arena(region, max((at + 1) // 2, 1))
    return arena[0], 2 * arena[3], base, at - base, lens, table.tobytes(), time.perf_counter() - t0


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
