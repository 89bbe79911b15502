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


def write_perturbed_pore_model(path: str) -> str:
    """A pore-model table in upstream's format (example/deps/template_median68pA.model: tab-separated, columns kmer and
    level_mean among others) whose levels differ from the built-in r9.4 table -- a deterministic function of it -- for the tests
    of the `pore_model_path` configuration key (tests/golden/cfg_keys.*: recorded from upstream with this very file)."""
    from warpstr_amd import pore_model
    level = np.load(pore_model._DATA)
    i = np.arange(len(level))
    new = level * 1.04 + ((i * 2654435761) % 1009) / 400.0 - 1.2   # (products, sums and one exact division: no libm)
    with open(path, 'w') as f:
        f.write('kmer\tlevel_mean\tlevel_stdv\n')
        for k, v in enumerate(new):
            kmer = ''.join('ACGT'[(k >> (2 * (5 - b))) & 3] for b in range(6))
            f.write(f'{kmer}\t{float(v)!r}\t1.5\n')
    return path
