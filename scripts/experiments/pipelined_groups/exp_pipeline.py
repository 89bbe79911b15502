"""The many_loci workload (2 000 loci x 30 reads from host memory) through main_wrapper_loci in one piece and in pipelined groups
(loci._pipelined), with the per-phase sums of the groups: where does a group's handle cost?"""
import json
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from warpstr_amd.wrapper import main_wrapper_loci

n_loci = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
root = tempfile.mkdtemp(prefix='wsx_pipe_', dir=bench.scratch_dir())
try:
    from bench import MANY_LOCI_PATTERNS, make_locus_dirs
    specs = [(f'locus{i:04d}', MANY_LOCI_PATTERNS[i % len(MANY_LOCI_PATTERNS)], 110, (2271, 3701), 5000 + i) for i in range(n_loci)]
    out = {}
    for tag, env in (('one_piece', '1'), ('pipelined', '')):
        loci, raws = make_locus_dirs(os.path.join(root, tag), specs, 30, 77)
        if tag == 'one_piece':
            main_wrapper_loci(loci[:8], 1, raw_reads=raws, quiet=True)
            loci, raws = make_locus_dirs(os.path.join(root, tag + '2'), specs, 30, 77)
        os.environ['WARPSTR_NO_PIPELINE'] = env
        if not env:
            del os.environ['WARPSTR_NO_PIPELINE']
        tm = {}
        t0 = time.perf_counter()
        main_wrapper_loci(loci, 16, raw_reads=raws, quiet=True, timings=tm)
        out[tag] = {'wall_s': time.perf_counter() - t0, **{k: v for k, v in tm.items() if isinstance(v, (int, float, dict, str)) and k not in ('loci_set_up',)}}
    print(json.dumps(out, indent=1, default=str))
finally:
    shutil.rmtree(root, ignore_errors=True)
