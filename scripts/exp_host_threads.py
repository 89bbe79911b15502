"""How the per-locus host work of main_wrapper_loci scales with host threads (native code without the GIL, csrc/host_loci.cpp):
N loci x 30 reads set up (overview.csv, flank file, two automata, state_similarity.csv) and their outputs written with a
stand-in for the GPU (every read gets a fixed record and sequence) on 1, 2, 4, 8, 16 threads.  No GPU needed.
Usage: exp_host_threads.py [n_loci]"""
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from warpstr_amd import _lib
from warpstr_amd.wrapper import main_wrapper_loci


class Stub:
    def __init__(self, tables, flank_lengths, caller_config, rescaler_config, device):
        pass

    def submit_raw(self, raws, lo, hi, aut):
        return len(raws)

    def collect(self, n):
        rec = np.zeros(n, dtype=_lib.RESULT_DTYPE)
        rec['len1'], rec['len2'], rec['cost1'], rec['cost2'] = 40, 44, 0.51234, 0.43219
        pos = np.arange(n + 1, dtype=np.int64)
        return rec, np.frombuffer(b'ACGT' * 10 * n, np.uint8), pos * 40, np.frombuffer(b'AAAT' * 11 * n, np.uint8), pos * 44

    def info(self):
        return {}

    def close(self):
        pass


def main():
    n_loci = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    root = tempfile.mkdtemp(prefix='wsx_threads_', dir=bench.scratch_dir())
    try:
        specs = [(f'locus{i:04d}', bench.MANY_LOCI_PATTERNS[i % 10], 110, (2271, 3701), 5000 + i) for i in range(n_loci)]
        for threads in (1, 2, 4, 8, 16):
            loci, raws = bench.make_locus_dirs(os.path.join(root, f't{threads}'), specs, 30, 77)
            tm = {}
            t0 = time.perf_counter()
            main_wrapper_loci(loci, threads, raw_reads=raws, quiet=True, timings=tm, _engine=Stub)
            dt = time.perf_counter() - t0
            print(f'{threads:2d} threads: {n_loci / dt:8.0f} loci/s  set-up {tm["setup_wall_s"] / n_loci * 1e3:.3f} ms/locus wall '
                  f'({tm["native_setup_s"] / n_loci * 1e3:.3f} in the library, summed over threads)  outputs {tm["store_s"] / n_loci * 1e3:.3f} ms/locus wall',
                  flush=True)
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == '__main__':
    main()
