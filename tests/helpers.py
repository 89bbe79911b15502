"""Shared helpers for the parity tests (oracle side)."""
import os

import numpy as np

from oracle import oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
DEFAULT_CASES = ['agc_fl16', 'agc_fl29', 'aaat_fl110', 'hd_fl20', 'dm2_fl40', 'ngc_fl20', 'agc_fl16_ragged', 'real_aaat']


def load_case(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def golden_automaton(z, tag):
    """Oracle automaton from the tables stored in a fixture (tag 't' template / 'r' reverse)."""
    return oracle.Automaton(z[f'{tag}_value'], z[f'{tag}_seq_idx'], z[f'{tag}_pred_ptr'], z[f'{tag}_pred_idx'],
                            z[f'{tag}_mask'], int(z[f'{tag}_endstate']), int(z['flank_length']))


def assert_close_rel(a, b, rel=1e-5):
    a, b = float(a), float(b)
    if np.isnan(a) and np.isnan(b):
        return
    if np.isinf(a) or np.isinf(b):
        assert a == b
        return
    assert abs(a - b) <= rel * max(abs(a), abs(b), 1e-300), (a, b)
