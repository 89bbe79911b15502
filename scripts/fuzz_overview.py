"""Differential sweep of the native overview.csv path (csrc/host_loci.cpp) against pandas on hostile CSV TEXT: tables whose cells are
drawn from tokens pandas treats specially (numbers in every spelling, NA strings, booleans in other cases, padded strings, out-of-range
exponents ...).  A table the native parser accepts must give the same columns to the caller and the same overview.csv after
store_results, byte for byte; most of these tables must be declined.  Usage: fuzz_overview.py [seed] [cases]; exit code 1 on a mismatch."""
import os, sys, tempfile, shutil
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from warpstr_amd import _hostlib, overview as ov
TOK = ['', '0', '1', '-1', '7', '12', '007', '-0', '+3', '3.0', '2.5', '-0.125', '0.1', '1e5', '1E5', '1e-05', '1e+16', '.5', '5.', '0.10', '1_0', 'inf', '-inf', 'Inf', 'nan', 'NaN', 'NA',
       'N/A', 'null', 'None', 'True', 'False', 'true', 'FALSE', 'abc', 'a b', ' x', 'x ', 'run_0', '0x1A', '1e400', '1e-400', '123456789012345678', '1234567890123456789', '12345678901234567890',
       '0.30000000000000004', '0.0015732835352270625', '3.141592653589793238462643', '-', '+', 'e5', '1e', '--1', '1.2.3', 'é', '#N/A', '<NA>', '1,5'.replace(',', ';'), "it's", 'a"b'.replace('"', "'")]
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
root = tempfile.mkdtemp(dir='/dev/shm' if os.path.isdir('/dev/shm') else None)
acc = dec = bad = 0
for case in range(n_cases):
    n = int(rng.integers(1, 9))
    extra = int(rng.integers(0, 4))
    cols = ['read_name', 'run_id', 'reverse', 'saved', 'l_start_raw', 'r_end_raw'] + [f'x{i}' for i in range(extra)]
    rows = []
    for r in range(n):
        row = {'read_name': f'r{r}', 'run_id': 'run_0', 'reverse': str(bool(rng.integers(0, 2))), 'saved': str(int(rng.integers(0, 2))),
               'l_start_raw': str(int(rng.integers(0, 100))), 'r_end_raw': str(int(rng.integers(200, 900)))}
        for c in cols[6:]:
            row[c] = ''
        rows.append(row)
    # poison: a few columns get tokens from a small per-column pool (so that columns are sometimes homogeneous)
    for c in rng.choice(cols, size=int(rng.integers(1, 4)), replace=False):
        pool = list(rng.choice(TOK, size=int(rng.integers(1, 4))))
        for row in rows:
            if rng.random() < 0.8:
                row[c] = str(rng.choice(pool))
    if not any(r['saved'] not in ('0', '', 'False') for r in rows):
        rows[0]['saved'] = '1'
    order = list(rng.permutation(cols)) if rng.random() < 0.3 else cols
    text = ','.join(order) + '\n' + ''.join(','.join(row[c] for c in order) + '\n' for row in rows)
    a, b = os.path.join(root, f'a{case}'), os.path.join(root, f'b{case}')
    for d in (a, b):
        os.makedirs(d)
        open(os.path.join(d, 'overview.csv'), 'w').write(text)
    nat = _hostlib.NativeOverview.open(os.path.join(a, 'overview.csv'))
    if nat is None:
        dec += 1
        continue
    acc += 1
    try:
        path, ref = ov.load_overview(b)
        saved = np.flatnonzero(np.asarray(ref['saved']).astype(bool))
        ok = (nat.saved.tolist() == saved.tolist() and nat.names == [str(x) for x in ref.index.to_numpy()[saved]]
              and nat.reverse.tolist() == np.asarray(ref['reverse'])[saved].astype(bool).tolist()
              and nat.lo.tolist() == np.asarray(ref['l_start_raw'])[saved].astype(np.int64).tolist()
              and nat.hi.tolist() == np.asarray(ref['r_end_raw'])[saved].astype(np.int64).tolist()
              and nat.run_id == [str(x) for x in np.asarray(ref['run_id'])[saved]])
        ns = len(saved)
        ov.store_results(path, ref, [('AC', 'ACG')] * ns, [(0.5, 0.25)] * ns, b, write=True)
        nat.store(a, [2] * ns, [3] * ns, [0.5] * ns, [0.25] * ns, np.frombuffer(b'ACG' * ns, np.uint8), np.arange(ns) * 3, write=True)
        same = open(os.path.join(a, 'overview.csv'), 'rb').read() == open(os.path.join(b, 'overview.csv'), 'rb').read()
    except Exception as e:
        ok, same = False, False
        print('pandas path raised', type(e).__name__, e)
    if not (ok and same):
        bad += 1
        print('MISMATCH case', case, 'ok', ok, 'same', same)
        print(text)
        if bad > 5: break
    shutil.rmtree(a); shutil.rmtree(b)
print('accepted', acc, 'declined', dec, 'mismatches', bad)
shutil.rmtree(root, ignore_errors=True)
sys.exit(1 if bad else 0)
