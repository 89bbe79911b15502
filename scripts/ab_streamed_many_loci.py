"""(Historic: the in-memory streamed run this script measured was taken out again -- profiles/r06_streamed_many_loci_ab.json.)
Same-box A/B of the streamed run on bench.py's many_loci workload (2 000 loci x 30 reads in host memory, 16 threads): with the
set-up, calling and handle growth as one pipeline, and with WARPSTR_NO_STREAMED_RUN (everything set up first)."""
import json
import os
import shutil
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from warpstr_amd.wrapper import main_wrapper_loci  # noqa: E402

n_loci = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
root = tempfile.mkdtemp(prefix='wsx_ab_', dir=bench.scratch_dir())
try:
    specs = [(f'locus{i:04d}', bench.MANY_LOCI_PATTERNS[i % len(bench.MANY_LOCI_PATTERNS)], 110, (2271, 3701), 5000 + i) for i in range(n_loci)]
    out = {'streamed': [], 'all_set_up_first': []}
    raws = None
    for r in range(6):
        tag = 'streamed' if r % 2 == 0 else 'all_set_up_first'
        loci, raws_r = bench.make_locus_dirs(os.path.join(root, f'run{r}'), specs, 30, 77)
        raws = raws or raws_r
        if tag == 'streamed':
            os.environ.pop('WARPSTR_NO_STREAMED_RUN', None)
        else:
            os.environ['WARPSTR_NO_STREAMED_RUN'] = '1'
        tm = {}
        main_wrapper_loci(loci, 16, raw_reads=raws, device=0, quiet=True, timings=tm)
        if r >= 2:   # (the first run of either kind warms the code objects and the staging)
            out[tag].append({'loci_per_s': round(n_loci / tm['total_s']), 'wall_s': round(tm['total_s'], 4), 'mode': tm.get('reader_mode'),
                             'setup_wall_s': round(tm['setup_wall_s'], 4), 'handle_s': round(tm['handle_s'], 4), 'read_s': round(tm['read_s'], 4),
                             'submit_s': round(tm['submit_s'], 4), 'collect_s': round(tm['collect_s'], 4), 'store_s': round(tm['store_s'], 4),
                             'batches': tm.get('batches')})
        shutil.rmtree(os.path.join(root, f'run{r}'), ignore_errors=True)
    print(json.dumps(out, indent=1))
finally:
    shutil.rmtree(root, ignore_errors=True)
