"""Live cross-check of the CPU oracle against the upstream caller itself, on inputs that are NOT in the fixtures.
Runs only where /root/reference exists (the development container); skipped everywhere else."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.reference

_CHILD = r'''
import json, sys
sys.path.insert(0, sys.argv[1] + '/tests/golden'); sys.path.insert(0, sys.argv[1])
import numpy as np
from _ref_import import import_reference
from warpstr_amd import synth
ns = import_reference()
pattern, fl, seed = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
locus = synth.make_locus(pattern, fl, seed)
sigs, revs, _ = synth.batch(locus, 3, (700, 1100), seed + 1, lo=3, hi=12)
rev_seq = locus.left_r + ns.wrapper.CallerWrapper.reverse_uniq_sequence(pattern) + locus.right_r
stas = {False: ns.automata.StateAutomata(locus.left_t + pattern + locus.right_t), True: ns.automata.StateAutomata(rev_seq)}
out = []
for s, r in zip(sigs, revs):
    sta = stas[r]
    w = ns.caller.WarpSTR(fl, sta.states, sta.endstate, sta.mask, None, r, 'x')
    try:
        res = w.run(s)
        out.append(dict(ok=True, len1=len(res.seq), len2=len(res.resc_seq), cost1=float(res.cost), cost2=float(res.resc_cost)))
    except Exception as e:
        out.append(dict(ok=False, err=type(e).__name__))
print('RESULT ' + json.dumps(out))
'''


@pytest.mark.parametrize('pattern,fl,seed', [('(AGC)', 16, 9001), ('(CTG)AA(CCG)', 18, 9002)])
def test_oracle_agrees_with_live_reference(pattern, fl, seed):
    if not os.path.isdir('/root/reference/src/caller'):
        pytest.skip('upstream reference not present on this machine')
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    p = subprocess.run([sys.executable, '-c', _CHILD, ROOT, pattern, str(fl), str(seed)], capture_output=True, text=True,
                       env=env, timeout=600)
    line = [l for l in p.stdout.splitlines() if l.startswith('RESULT ')]
    assert line, p.stderr[-2000:]
    ref = json.loads(line[0][7:])
    sys.path.insert(0, ROOT)
    from oracle import oracle
    from warpstr_amd import synth
    locus = synth.make_locus(pattern, fl, seed)
    sigs, revs, _ = synth.batch(locus, 3, (700, 1100), seed + 1, lo=3, hi=12)
    oa = {False: oracle.Automaton.from_table(locus.template, fl), True: oracle.Automaton.from_table(locus.reverse, fl)}
    for s, r, exp in zip(sigs, revs, ref):
        o = oracle.call_read(oa[r], s)
        if exp['ok']:
            assert o.status == 0 and (o.len1, o.len2) == (exp['len1'], exp['len2'])
            assert abs(o.cost1 - exp['cost1']) <= 1e-12 * abs(exp['cost1'])
            assert abs(o.cost2 - exp['cost2']) <= 1e-12 * abs(exp['cost2'])
        else:
            assert o.status != 0
