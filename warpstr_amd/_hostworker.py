"""A reader process of main_wrapper_loci (warpstr_amd/loci.py: _WorkerPool): reads pickled (function name, argument) pairs from
its standard input, runs the named function of warpstr_amd._readers (the fast5 files of a batch), writes the pickled
('ok', result) or ('err', text) to its standard output.  Started with `python -m warpstr_amd._hostworker K` (or forked, sixteen at
a time, by `python -m warpstr_amd._hostworker --fork ...`: fork_readers), so it never imports
the parent's main module (multiprocessing's spawn would), never touches HIP, and imports the fast5 reader's NumPy-free core only
(NumPy itself when a function that returns arrays is asked for)."""
import os
import pickle
import sys
import traceback


def serve(src, out):
    """The loop of a reader: tasks from `src`, answers to `out` (binary file objects)."""
    from warpstr_amd import _readers
    names = {'_read_chunk': _readers.read_chunk, '_probe_chunk': _readers.probe_chunk, '_decode_chunk': _readers.decode_chunk,
             'read_chunk': _readers.read_chunk, 'probe_chunk': _readers.probe_chunk, 'decode_chunk': _readers.decode_chunk,
             'decode_arena': _readers.decode_arena, 'pack_arena': _readers.pack_arena,
             'loaded_modules': lambda prefix: sorted(m for m in sys.modules if m.startswith(prefix))}   # (what a test asks)
    while True:
        try:
            name, arg = pickle.load(src)
        except EOFError:
            return
        try:
            res = ('ok', names[name](arg))
        except Exception:  # noqa: BLE001 -- reported to the parent, which raises
            res = ('err', traceback.format_exc())
        pickle.dump(res, out, protocol=pickle.HIGHEST_PROTOCOL)
        out.flush()


def main():
    if len(sys.argv) > 2 and sys.argv[1] == '--fork':
        return fork_readers([int(x) for x in sys.argv[2:]])
    out = os.fdopen(os.dup(sys.stdout.fileno()), 'wb')  # results go here; whatever the functions print goes to stderr
    os.dup2(sys.stderr.fileno(), sys.stdout.fileno())
    from warpstr_amd import _readers
    if len(sys.argv) > 1 and sys.argv[1].isdigit():
        _readers.spread_over_cpus(int(sys.argv[1]))   # (the k-th worker starts on the k-th CPU of the mask; nothing stays pinned)
    serve(sys.stdin.buffer, out)


def fork_readers(fds):
    """`python -m warpstr_amd._hostworker --fork r0 w0 r1 w1 ...`: ONE interpreter started by the parent -- whose address space,
    with the GPU runtime mapped, makes every process it starts itself cost ~10 ms -- imports what a reader needs and forks a
    reader per pair of pipe ends (tasks in, answers out).  It prints the readers' process numbers, one line, and stays until they
    have all ended.  Sixteen readers are up in the time of one interpreter's start."""
    from warpstr_amd import _h5core, _readers  # noqa: F401 -- imported once, inherited by every reader (no library is loaded yet)
    pairs = list(zip(fds[0::2], fds[1::2]))
    os.dup2(sys.stderr.fileno(), 1)            # (what a reader's functions print goes to stderr)
    pids = []
    for k, (r, w) in enumerate(pairs):
        pid = os.fork()
        if pid == 0:
            code = 1   # (a reader that dies on an exception says so with its exit code)
            try:
                for r2, w2 in pairs:
                    if r2 != r:
                        os.close(r2)
                        os.close(w2)
                os.close(PID_FD)
                _readers.spread_over_cpus(k + 1)
                if WARM_REGIONS:
                    _readers.prepare_arenas(range(WARM_REGIONS))
                serve(os.fdopen(r, 'rb'), os.fdopen(w, 'wb'))
                code = 0
            except BaseException:  # noqa: BLE001
                traceback.print_exc()
            finally:
                try:
                    _readers._drop_arenas()   # (os._exit runs no atexit hook: the arenas' memory-backed files go here)
                finally:
                    os._exit(code)
        pids.append(pid)
    for r, w in pairs:
        os.close(r)
        os.close(w)
    os.write(PID_FD, (' '.join(str(p) for p in pids) + '\n').encode())
    os.close(PID_FD)
    for p in pids:
        try:
            os.waitpid(p, 0)
        except ChildProcessError:
            pass


PID_FD = None
WARM_REGIONS = int(os.environ.get('WARPSTR_WARM_ARENAS', '0') or 0)   # regions a forked reader creates and touches before its first task

if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--fork':
        PID_FD = os.dup(1)   # the parent reads the readers' process numbers here (taken before anything redirects stdout)
    main()
