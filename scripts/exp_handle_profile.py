"""Why the host side of the handle's creation took 0.48 s beside reader processes and 0.02 s without: the from_fast5 leg with
HipCaller.__init__ under cProfile, wall against the thread's CPU time, and the threads alive at that moment."""
import cProfile
import io
import json
import os
import pstats
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from warpstr_amd import caller

orig = caller.HipCaller.__init__


def wrapped(self, *a, **k):
    pr = cProfile.Profile()
    w0, c0 = time.perf_counter(), time.thread_time()
    pr.enable()
    try:
        orig(self, *a, **k)
    finally:
        pr.disable()
        w1, c1 = time.perf_counter(), time.thread_time()
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(12)
        print(f'HipCaller.__init__: wall {w1 - w0:.3f} s, CPU of this thread {c1 - c0:.3f} s, threads: '
              f'{sorted(t.name for t in threading.enumerate())}\n{s.getvalue()}', file=sys.stderr, flush=True)


caller.HipCaller.__init__ = wrapped
print(json.dumps(bench.from_fast5_leg(int(sys.argv[1]) if len(sys.argv) > 1 else 3000, 0), indent=1))
