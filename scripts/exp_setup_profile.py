"""Where the per-locus set-up of main_wrapper_loci spends its time on the host (no GPU): N loci of the upstream test locus, the native
part (overview, flanks, automata: _host_loci.so, without the GIL) against the Python around it (LocusJob), one thread, cProfile.
Usage: exp_setup_profile.py [n_loci]"""
import cProfile
import io
import json
import os
import pstats
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pandas as pd

from warpstr_amd import _hostlib, loci as L, overview as ov
from warpstr_amd.caller import CallerConfig
from warpstr_amd.pore_model import default_pore_model
from warpstr_amd.wrapper import LocusPath

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
real = os.path.join(ROOT, 'tests', 'golden', 'real')
fj = json.load(open(os.path.join(real, 'flanks.json')))
ex = pd.read_csv(os.path.join(real, 'example.csv'), dtype={'read_name': str})
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
root = tempfile.mkdtemp(prefix='wsx_setup_', dir='/dev/shm' if os.path.isdir('/dev/shm') else None)
try:
    head = 'read_name,fast5_path,reverse,l_start_raw,r_end_raw,run_id,saved\n'
    rows = [f'{nm},/nowhere/batch.fast5,{bool(rv)},{int(a)},{int(b)},run_0,1\n' for nm, rv, a, b in
            zip(ex['read_name'], ex['reverse'].astype(bool), ex['l_start_raw'], ex['r_end_raw'])]
    flanks = [fj['left_template'], fj['right_template'], fj['left_reverse'], fj['right_reverse']]
    loci = []
    for i in range(n):
        loc = os.path.join(root, f'copy{i:05d}')
        ov.store_flanks(loc, flanks)
        with open(os.path.join(loc, 'overview.csv'), 'w') as f:
            f.writelines([head] + rows)
        loci.append(LocusPath(loc, fj['sequence'], int(fj['flank_length']), f'copy{i:05d}'))
    pm, cc = default_pore_model(), CallerConfig()

    def setup(chunk):
        t1 = time.perf_counter()
        sts = _hostlib.NativeSetup.run_many([l.path for l in chunk], [l.sequence.upper() for l in chunk], pm, cc.min_state_similarity, True)
        t2 = time.perf_counter()
        ptm = {}
        jobs = [L.LocusJob(l, pm, ptm, cc, write=True, native=True, setup=sts[q]) for q, l in enumerate(chunk)]
        return jobs, t2 - t1, time.perf_counter() - t2

    setup(loci[:64])
    # the library call by itself (one thread): what is left of a part's time is Python
    h = _hostlib.lib()
    real, spent = h.wsh_loci_setup, [0.0]

    def timed(*a):
        t = time.perf_counter()
        try:
            return real(*a)
        finally:
            spent[0] += time.perf_counter() - t
    h.wsh_loci_setup = timed
    t0 = time.perf_counter()
    for k in range(64, n, 64):
        setup(loci[k:k + 64])
    wall = time.perf_counter() - t0
    h.wsh_loci_setup = real
    print(json.dumps({'threads': 1, 'loci': n - 64, 'wall_s': wall, 'inside_wsh_loci_setup_s': spent[0], 'python_us_per_locus': (wall - spent[0]) / (n - 64) * 1e6,
                      'native_us_per_locus': spent[0] / (n - 64) * 1e6}))
    # the same on eight threads, as a run does it: the native part side by side, the Python part one thread at a time -- the wall-clock
    # is about the Python part's
    from concurrent.futures import ThreadPoolExecutor
    for _ in range(3):
        with ThreadPoolExecutor(8, initializer=L.spread_over_cpus) as ex:
            t0 = time.perf_counter()
            list(ex.map(setup, [loci[k:k + 64] for k in range(64, n, 64)]))
            print(json.dumps({'threads': 8, 'loci': n - 64, 'wall_s': time.perf_counter() - t0, 'us_per_locus': (time.perf_counter() - t0) / (n - 64) * 1e6}))
    pr = cProfile.Profile()
    nat = py = 0.0
    t0 = time.perf_counter()
    pr.enable()
    for k in range(64, n, 64):
        _, a, b = setup(loci[k:k + 64])
        nat, py = nat + a, py + b
    pr.disable()
    wall = time.perf_counter() - t0
    print(json.dumps({'loci': n - 64, 'wall_s': wall, 'native_s': nat, 'python_s': py, 'python_us_per_locus': py / (n - 64) * 1e6,
                      'native_us_per_locus': nat / (n - 64) * 1e6}))
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats('tottime').print_stats(18)
    print(out.getvalue())
finally:
    shutil.rmtree(root, ignore_errors=True)
