"""The many_loci workload (2 000 loci x 30 reads from host memory, 16 threads) under cProfile: where the calling thread spends the
run, phase sums from the timings."""
import cProfile
import io
import json
import os
import pstats
import shutil
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bench import MANY_LOCI_PATTERNS, make_locus_dirs
from warpstr_amd.wrapper import main_wrapper_loci

n_loci = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
root = tempfile.mkdtemp(prefix='wsx_prof_', dir=bench.scratch_dir())
try:
    specs = [(f'locus{i:04d}', MANY_LOCI_PATTERNS[i % len(MANY_LOCI_PATTERNS)], 110, (2271, 3701), 5000 + i) for i in range(n_loci)]
    warm, raws = make_locus_dirs(os.path.join(root, 'warm'), specs[:16], 30, 77)
    main_wrapper_loci(warm, 1, raw_reads=raws, quiet=True)
    for rep in range(2):
        loci, raws = make_locus_dirs(os.path.join(root, f'run{rep}'), specs, 30, 77)
        tm = {}
        pr = cProfile.Profile()
        pr.enable()
        main_wrapper_loci(loci, threads, raw_reads=raws, quiet=True, timings=tm)
        pr.disable()
        out = io.StringIO()
        pstats.Stats(pr, stream=out).sort_stats('tottime').print_stats(22)
        print(json.dumps({k: v for k, v in tm.items() if isinstance(v, (int, float))}, indent=0)[:1500])
        print(out.getvalue()[:4500])
finally:
    shutil.rmtree(root, ignore_errors=True)
