"""How long the interpreters of the fast5 reader processes take to come up on this box: bare Python, ctypes + libhdf5 + libzstd, NumPy,
the reader module -- one at a time and sixteen at once."""
import glob
import subprocess
import sys
import time

H5 = "import ctypes, glob, mmap, pickle; ctypes.CDLL(sorted(glob.glob('/opt/conda/lib/libhdf5.so*') + glob.glob('/usr/lib/*/libhdf5*.so*'))[0])"
CASES = [('python -c pass', 'pass'), ('ctypes + mmap + pickle', 'import ctypes, mmap, pickle, os, sys'), ('... + libhdf5', H5),
         ('import numpy', 'import numpy'), ('warpstr_amd._readers + fast5 libs', 'from warpstr_amd import _readers, fast5; fast5._libs()')]
for name, code in CASES:
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        subprocess.run([sys.executable, '-c', code], check=True)
        best = min(best, time.perf_counter() - t0)
    print(f'{name:36s} {best * 1e3:7.1f} ms', flush=True)
for name, code in (CASES[2], CASES[3], CASES[4]):
    t0 = time.perf_counter()
    ps = [subprocess.Popen([sys.executable, '-c', code]) for _ in range(16)]
    for p in ps:
        p.wait()
    print(f'16 at once: {name:24s} {(time.perf_counter() - t0) * 1e3:7.1f} ms', flush=True)
