"""Debug / measurement aid for bench.py's many_loci leg: L loci through one handle and through one handle per locus, which files differ."""
import filecmp
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from warpstr_amd.wrapper import main_wrapper, main_wrapper_loci  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n_loop = int(sys.argv[2]) if len(sys.argv) > 2 else L
root = tempfile.mkdtemp(prefix='wsx_ml_')
specs = [(f'locus{i:04d}', bench.MANY_LOCI_PATTERNS[i % len(bench.MANY_LOCI_PATTERNS)], 110, (2271, 3701), 5000 + i) for i in range(L)]
loci, raws = bench.make_locus_dirs(os.path.join(root, 'batched'), specs, 30, 77)
loop, _ = bench.make_locus_dirs(os.path.join(root, 'loop'), specs[:n_loop], 30, 77)
reader = lambda path: raws[os.path.basename(path)[:-len('.fast5')]]
tm = {}
t0 = time.perf_counter()
main_wrapper_loci(loci, 1, raw_reader=reader, quiet=True, timings=tm)
print('batched', time.perf_counter() - t0, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in tm.items()})
import contextlib, io
t0 = time.perf_counter()
with contextlib.redirect_stdout(io.StringIO()):
    for l in loop:
        main_wrapper(l, 1, raw_reader=reader)
print('loop', (time.perf_counter() - t0) / n_loop * 1e3, 'ms per locus')
import pandas as pd
for a, b in zip(loci, loop):
    for rel in ('overview.csv', 'predictions/sequences/all.fasta', 'summaries/state_similarity.csv'):
        if not filecmp.cmp(os.path.join(a.path, rel), os.path.join(b.path, rel), shallow=False):
            print('DIFF', a.name, a.sequence, rel)
            if rel == 'overview.csv':
                da, db = pd.read_csv(os.path.join(a.path, rel)), pd.read_csv(os.path.join(b.path, rel))
                for c in da.columns:
                    if not da[c].equals(db[c]):
                        bad = np.flatnonzero(~(da[c] == db[c]).to_numpy())
                        print('   column', c, 'rows', bad[:5], da[c].to_numpy()[bad[:3]], db[c].to_numpy()[bad[:3]])
