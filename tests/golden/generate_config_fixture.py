#!/usr/bin/env python3
"""Fixture for the configuration keys a WarpSTR YAML carries into step 3 / step 4 (`pore_model_path`,
`genotyping_config.{min_weight,std_filter}`): tests/golden/cfg_keys.npz + cfg_keys.json.

DEVELOPMENT-CONTAINER ONLY (needs /root/reference; see _ref_import.py).  One upstream process with a PERTURBED pore-model
table (tests/helpers.py: write_perturbed_pore_model -- the tests write the same file) and non-default genotyping settings:
  * the automata of a locus as upstream's StateAutomata builds them from that table (src/caller/automata.py:43-48,
    src/squiggler/pore_model.py:15-33),
  * WarpSTR.run on seeded reads against those automata (src/caller/caller.py:117-149),
  * run_genotyping_overview (src/genotyper/genotyping.py:68-82) on the overview those calls give, numpy's generator seeded.
Data only: arrays and strings, no reference source text.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/generate_config_fixture.py
"""
import contextlib
import io
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from _ref_import import REFERENCE_ROOT, import_reference  # noqa: E402
from generate_golden import write_case  # noqa: E402
from helpers import write_perturbed_pore_model  # noqa: E402

from warpstr_amd import synth  # noqa: E402
from warpstr_amd.pore_model import PoreModel  # noqa: E402

GENOTYPING = dict(min_weight=0.35, std_filter=1.5, visualize=False, msa=False)
PATTERN, FLANK, FSEED, N_READS, RSEED = '(AGC)AACAGCCGCCAC(CGC)', 20, 31, 12, 301


def main():
    tmp = tempfile.mkdtemp(prefix='warpstr_cfgkeys_')
    model = write_perturbed_pore_model(os.path.join(tmp, 'perturbed.model'))
    ns = import_reference(flank_length=FLANK, pore_model_path=model, genotyping_config=GENOTYPING)
    # the reads follow THIS table's levels (two alleles), so that the calls are sensible under it
    pm = PoreModel(model)
    locus = synth.make_locus(PATTERN, FLANK, FSEED, pore_model=pm)
    rng = np.random.default_rng(RSEED)
    sigs, revs, truth = [], [], []
    for i in range(N_READS):
        rev = bool(i & 1)
        reps = 9 if i % 3 else 17
        s, t = synth.squiggle(locus, rev, 1800, rng, lo=reps, hi=reps, sigma=0.2, pore_model=pm)
        sigs.append(s)
        revs.append(rev)
        truth.append(t)
    write_case(ns, 'cfg_keys', PATTERN, FLANK, locus, sigs, revs, truth, HERE)
    z = np.load(os.path.join(HERE, 'cfg_keys.npz'))
    lens = [len(str(z[f'r{i}_seq'][1])) for i in range(N_READS)]
    # step 4 on those calls with the non-default settings (upstream's own functions; numpy's generator seeded)
    import pandas as pd
    old = os.getcwd()
    os.chdir(REFERENCE_ROOT)
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        from src.genotyper import genotyping
    finally:
        os.chdir(old)
        sys.path.remove(REFERENCE_ROOT)
    assert genotyping.genotyping_config.min_weight == GENOTYPING['min_weight'] and genotyping.genotyping_config.std_filter == GENOTYPING['std_filter']
    out = dict(genotyping_config=GENOTYPING, pattern=PATTERN, flank_length=FLANK, lengths=lens, cases=[])
    extra = {'calls': lens,
             # the same settings on length sets where they matter: a light second component (0.2 < weight < 0.35 -> homozygous under
             # 0.35, heterozygous under the default), outliers between 1.5 and 2 standard deviations
             'light_component': [30] * 14 + [41] * 5, 'outliers': [25] * 9 + [26] * 8 + [31, 19]}
    for name, vals in extra.items():
        df = pd.DataFrame({'read_name': [f'read{i:03d}' for i in range(len(vals))], 'saved': True, 'results': vals}).set_index('read_name')
        loc = tempfile.mkdtemp(prefix='warpstr_cfgkeys_gt_')
        os.makedirs(os.path.join(loc, 'predictions'))
        buf = io.StringIO()
        np.random.seed(77)
        with contextlib.redirect_stdout(buf):
            genotyping.run_genotyping_overview(df, loc, None)
        out['cases'].append(dict(name=name, seed=77, results=[int(v) for v in vals], alleles_csv=open(os.path.join(loc, 'predictions', 'alleles.csv')).read(),
                                 stdout=buf.getvalue()))
        print(name, out['cases'][-1]['stdout'].strip())
    with open(os.path.join(HERE, 'cfg_keys.json'), 'w') as f:
        json.dump(out, f, indent=1)


if __name__ == '__main__':
    main()
