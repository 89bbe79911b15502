"""A reader process of main_wrapper_loci (warpstr_amd/loci.py: _WorkerPool): reads pickled (function name, argument) pairs from
its standard input, runs the named function of warpstr_amd._readers (the fast5 files of a batch), writes the pickled
('ok', result) or ('err', text) to its standard output.  Started with `python -m warpstr_amd._hostworker K`, so it never imports
the parent's main module (multiprocessing's spawn would), never touches HIP, and imports the fast5 reader's NumPy-free core only
(NumPy itself when a function that returns arrays is asked for)."""
import os
import pickle
import sys
import traceback


def main():
    out = os.fdopen(os.dup(sys.stdout.fileno()), 'wb')  # results go here; whatever the functions print goes to stderr
    os.dup2(sys.stderr.fileno(), sys.stdout.fileno())
    src = sys.stdin.buffer
    from warpstr_amd import _readers
    if len(sys.argv) > 1 and sys.argv[1].isdigit():
        _readers.spread_over_cpus(int(sys.argv[1]))   # (the k-th worker starts on the k-th CPU of the mask; nothing stays pinned)
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


if __name__ == '__main__':
    main()
