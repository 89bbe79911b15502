"""The configuration keys a WarpSTR YAML carries into steps 3 and 4 are honoured by `python -m warpstr_amd cfg.yaml`
(warpstr_amd/config.py, wrapper.main): `pore_model_path`, `genotyping_config.{min_weight,std_filter}`, `force_overwrite`,
`verbose`, and step 4 alone on an earlier run's overview.csv -- against tests/golden/cfg_keys.*, recorded from upstream running
with the same perturbed pore-model table and the same non-default genotyping settings (generate_config_fixture.py)."""
import json
import os

import numpy as np
import pandas as pd
import pytest
import yaml

from tests.helpers import GOLDEN, load_case, write_perturbed_pore_model
from warpstr_amd import overview as ov
from warpstr_amd.automata import locus_automata
from warpstr_amd.config import load_config
from warpstr_amd.wrapper import main, prepare_subdirs

FIX = json.load(open(os.path.join(GOLDEN, 'cfg_keys.json')))


def _config(tmp_path, **over):
    cfg = {'output': str(tmp_path / 'out'), 'reference_path': 'none', 'threads': 1, 'single_read_extraction': False,
           'guppy_annotation': False, 'exp_signal_generation': False, 'tr_region_extraction': False, 'tr_region_calling': False,
           'genotyping': True, 'flank_length': FIX['flank_length'],
           'tr_calling_config': {'visualize_alignment': False, 'visualize_phase': False, 'visualize_strand': False, 'visualize_cost': False},
           'genotyping_config': dict(FIX['genotyping_config']),
           'loci': [{'name': 'L', 'coord': 'chr1:1-2', 'sequence': FIX['pattern']}]}
    cfg.update(over)
    path = str(tmp_path / 'cfg.yaml')
    with open(path, 'w') as f:
        yaml.safe_dump(cfg, f)
    os.makedirs(tmp_path / 'out' / 'L', exist_ok=True)
    return path


def test_pore_model_path_gives_upstreams_automata(tmp_path):
    """The perturbed table through `pore_model_path`: state levels equal the ones upstream's StateAutomata built from it."""
    model = write_perturbed_pore_model(str(tmp_path / 'perturbed.model'))
    cfg = load_config(_config(tmp_path, pore_model_path=model))
    z = load_case('cfg_keys')
    flanks = [str(x) for x in z['flanks']]
    for table, tag in zip(locus_automata(*flanks, str(z['pattern']), cfg.pore_model()), 'tr'):
        assert np.array_equal(table.value, z[f'{tag}_value'])
        assert np.array_equal(table.pred_idx, z[f'{tag}_pred_idx'])
    default = load_config(_config(tmp_path))   # upstream's default path, not present here: the same table from the package's data
    assert not np.array_equal(locus_automata(*flanks, str(z['pattern']), default.pore_model())[0].value, z['t_value'])
    with pytest.raises(FileNotFoundError):
        load_config(_config(tmp_path, pore_model_path=str(tmp_path / 'nowhere.model'))).pore_model()


@pytest.mark.parametrize('case', FIX['cases'], ids=[c['name'] for c in FIX['cases']])
def test_genotyping_alone_with_the_configured_settings(tmp_path, capsys, case):
    """tr_region_calling: False, genotyping: True -- step 4 runs from the overview.csv of an earlier run (WarpSTR.py:71-79) with
    genotyping_config's min_weight / std_filter: alleles.csv and the printed summary as upstream wrote them."""
    path = _config(tmp_path)
    loc = tmp_path / 'out' / 'L'
    n = len(case['results'])
    pd.DataFrame({'read_name': [f'read{i:03d}' for i in range(n)], 'saved': True, 'results': case['results']}).to_csv(loc / 'overview.csv', index=False)
    np.random.seed(case['seed'])
    main(['--config', path])
    assert open(loc / 'predictions' / 'alleles.csv').read() == case['alleles_csv']
    assert case['stdout'].strip() in capsys.readouterr().out


def test_the_settings_change_the_genotype(tmp_path, capsys):
    """The same calls under the DEFAULT genotyping settings give the other answer: the keys are not decoration."""
    case = FIX['cases'][0]
    path = _config(tmp_path, genotyping_config={'visualize': False})
    loc = tmp_path / 'out' / 'L'
    n = len(case['results'])
    pd.DataFrame({'read_name': [f'read{i:03d}' for i in range(n)], 'saved': True, 'results': case['results']}).to_csv(loc / 'overview.csv', index=False)
    np.random.seed(case['seed'])
    main(['--config', path])
    assert open(loc / 'predictions' / 'alleles.csv').read() != case['alleles_csv']
    assert '(66, 114)' in capsys.readouterr().out


def test_what_cannot_be_honoured_is_refused_or_reported(tmp_path, capsys):
    with pytest.raises(ValueError, match='msa'):
        load_config(_config(tmp_path, genotyping_config={'msa': True}))
    with pytest.raises(AssertionError):
        load_config(_config(tmp_path, genotyping_config={'min_weight': 1.5}))
    cfg = load_config(_config(tmp_path, tr_region_calling=True, tr_calling_config={'visualize_cost': True}, genotyping_config={'visualize': True}))
    notes = cfg.notices()
    assert any('visualize_cost' in n for n in notes) and any('alleles.svg' in n for n in notes)


def test_force_overwrite_and_verbose(tmp_path, capsys):
    """force_overwrite empties the step's directories before the run (src/helpers.py:32-69), nothing else; verbose prints the
    durations (src/helpers.py:16-29)."""
    loc = tmp_path / 'out' / 'L'
    os.makedirs(loc / 'predictions' / 'sequences')
    os.makedirs(loc / 'expected_signals')
    (loc / 'predictions' / 'sequences' / 'stale.fasta').write_text('>x\n')
    (loc / 'predictions' / 'alleles.csv').write_text('old')
    (loc / 'expected_signals' / 'sequences.csv').write_text('kept')
    prepare_subdirs(str(loc), True, False)
    assert (loc / 'predictions' / 'sequences' / 'stale.fasta').exists() and (loc / 'predictions' / 'DTW_alignments').is_dir()
    prepare_subdirs(str(loc), True, True)
    assert not (loc / 'predictions' / 'sequences' / 'stale.fasta').exists() and not (loc / 'predictions' / 'alleles.csv').exists()
    assert (loc / 'predictions' / 'basecalls').is_dir() and (loc / 'summaries').is_dir()
    assert (loc / 'expected_signals' / 'sequences.csv').read_text() == 'kept'
    case = FIX['cases'][2]
    path = _config(tmp_path, verbose=1, force_overwrite=True)
    pd.DataFrame({'read_name': [f'r{i}' for i in range(len(case['results']))], 'saved': True, 'results': case['results']}).to_csv(loc / 'overview.csv', index=False)
    np.random.seed(case['seed'])
    main(['--config', path])
    out = capsys.readouterr().out
    assert 'Duration for genotyping: 00h 00m' in out and 'Duration for whole' in out
    assert open(loc / 'predictions' / 'alleles.csv').read() == case['alleles_csv']   # (calling is off: its directories were left alone)
    assert open(loc / 'sequence.txt').read() == FIX['pattern']
