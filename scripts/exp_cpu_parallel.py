"""How many threads of plain native work this box runs at once: the same fixed loop (automaton compiles through
csrc/host_loci.cpp, no files, no Python between the calls) on 1, 2, 4, 8, 16 threads.  Tells apart "the host library does not
scale" from "the box does not have the cores"."""
import ctypes as C
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from warpstr_amd import _hostlib
from warpstr_amd.pore_model import default_pore_model

pm = default_pore_model()
rng = np.random.default_rng(0)
pats = [(''.join('ACGT'[i] for i in rng.integers(0, 4, 110)) + p + ''.join('ACGT'[i] for i in rng.integers(0, 4, 110))).encode()
        for p in ['(AAAT)', '(AGC)AACAGCCGCCAC(CGC)', '((CAGG){CAGM})(CAGA)(CA)'] * 20]
n = len(pats)
arr_p = (C.c_char_p * n)(*pats)
h = _hostlib.lib()
lev = pm.level_norm.ctypes.data


def work(reps):
    a = _hostlib.WshAutomaton()
    for _ in range(reps):
        for p in pats:
            h.wsh_automaton_compile(p, len(p), lev, 6, C.byref(a))
            h.wsh_automaton_free(C.byref(a))


if '--worker' in sys.argv:
    work(200)
    sys.exit(0)
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
for nt in (1, 2, 4, 8, 16):
    ths = [threading.Thread(target=work, args=(40,)) for _ in range(nt)]
    t = time.perf_counter()
    [x.start() for x in ths]
    [x.join() for x in ths]
    dt = time.perf_counter() - t
    print(f'{nt:2d} threads: {nt * 40 * n / dt:9.0f} compiles/s', flush=True)
# the same as PROCESSES (no shared interpreter at all)
if '--worker' not in sys.argv:
    import subprocess
    for nt in (1, 4, 16):
        t = time.perf_counter()
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--worker'], stdout=subprocess.DEVNULL) for _ in range(nt)]
        [q.wait() for q in ps]
        print(f'{nt:2d} processes (start-up included): {time.perf_counter() - t:.2f} s for {nt} x 5 threads-worth of work', flush=True)
