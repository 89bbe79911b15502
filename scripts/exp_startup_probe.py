"""How long the interpreters of the fast5 reader processes take to come up on this box: bare Python, ctypes + libhdf5 + libzstd, NumPy,
the reader module with the array reader (what a reader process imported until round 5's last cycles) and with the NumPy-free core
(what it imports now: warpstr_amd/_h5core.py, library paths handed down by the parent) -- one at a time and sixteen at once."""
import glob
import os
import subprocess
import sys
import time

H5 = "import ctypes, glob, mmap, pickle; ctypes.CDLL(sorted(glob.glob('/opt/conda/lib/libhdf5.so*') + glob.glob('/usr/lib/*/libhdf5*.so*'))[0])"
CASES = [('python -c pass', 'pass'), ('ctypes + mmap + pickle', 'import ctypes, mmap, pickle, os, sys'), ('... + libhdf5', H5),
         ('import numpy', 'import numpy'), ('warpstr_amd._readers + fast5 libs', 'from warpstr_amd import _readers, fast5; fast5._libs()'),
         ('warpstr_amd._readers + _h5core libs', 'import pickle, traceback; from warpstr_amd import _readers, _h5core; _h5core.libs()')]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from warpstr_amd import _h5core  # noqa: E402

os.environ['WARPSTR_LIBHDF5'], os.environ['WARPSTR_LIBZSTD'] = _h5core.lib_paths()   # (as loci._WorkerPool hands them down)
os.environ['PYTHONPATH'] = sys.path[0] + os.pathsep + os.environ.get('PYTHONPATH', '')
for name, code in CASES:
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        subprocess.run([sys.executable, '-c', code], check=True)
        best = min(best, time.perf_counter() - t0)
    print(f'{name:36s} {best * 1e3:7.1f} ms', flush=True)
for name, code in CASES[2:]:
    t0 = time.perf_counter()
    ps = [subprocess.Popen([sys.executable, '-c', code]) for _ in range(16)]
    for p in ps:
        p.wait()
    print(f'16 at once: {name:36s} {(time.perf_counter() - t0) * 1e3:7.1f} ms', flush=True)
