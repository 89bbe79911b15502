"""Step 4 (warpstr_amd/genotyper.py) against what upstream's own genotyper produced on the same seeded inputs
(tests/golden/genotype.json, written by tests/golden/generate_golden.py --only genotype through the imported reference:
src/genotyper/genotyping.py run_genotyping / run_genotyping_overview / run_genotyping_complex).  scikit-learn's mixture
draws from numpy's global generator, so every case re-seeds it exactly as the generator did."""
import contextlib
import io
import json
import os

import numpy as np
import pandas as pd
import pytest

from warpstr_amd import genotyper

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, 'golden', 'genotype.json')) as _f:
    GOLD = json.load(_f)
SETTINGS = dict(min_weight=GOLD['min_weight'], std_filter=GOLD['std_filter'])


@pytest.mark.parametrize('case', GOLD['simple'], ids=[c['name'] for c in GOLD['simple']])
def test_simple_repeat_genotype_matches_upstream(case):
    np.random.seed(case['seed'])
    call = genotyper.call_alleles(case['values'], **SETTINGS)
    assert [list(c) for c in call.clusters] == [g for g in (case['group1'], case['group2']) if g or g is case['group1']]
    assert [int(v) for v in call.labels] == case['predictions']
    assert [call.allele(0), call.allele(1)] == case['alleles']
    assert [call.support(0), call.support(1)] == case['sizes']


def test_the_fixture_covers_both_outcomes_and_the_small_sets():
    names = {c['name']: c for c in GOLD['simple']}
    assert sum(1 for c in GOLD['simple'] if c['group2']) >= 3 and sum(1 for c in GOLD['simple'] if not c['group2']) >= 5
    assert len(names['five_values_no_filter']['group1']) == 5             # five reads or fewer: nothing is set aside
    assert len(names['outliers_filtered']['group1']) < len(names['outliers_filtered']['values'])
    assert names['minor_component_below_min_weight']['alleles'][1] == '-'  # a component lighter than min_weight: one allele


@pytest.mark.parametrize('case', GOLD['overview'], ids=[c['name'] for c in GOLD['overview']])
def test_overview_genotype_and_alleles_csv_match_upstream(case, tmp_path):
    """run_genotyping_overview(overview, locus_path, muscle_path): rows that were not `saved` are skipped, basecalled lengths
    r_seq_start - l_seq_end are genotyped when the overview has them, alleles.csv is byte-identical, so is what is printed."""
    df = pd.DataFrame(case['columns']).set_index('read_name')
    buf = io.StringIO()
    np.random.seed(case['seed'])
    with contextlib.redirect_stdout(buf):
        genotyper.run_genotyping_overview(df, str(tmp_path), None, **SETTINGS)
    assert (tmp_path / 'predictions' / 'alleles.csv').read_text() == case['alleles_csv']
    assert buf.getvalue() == case['stdout']


def test_overview_is_read_from_the_locus_directory_when_not_given(tmp_path):
    case = GOLD['overview'][1]
    pd.DataFrame(case['columns']).to_csv(tmp_path / 'overview.csv', index=False)
    np.random.seed(case['seed'])
    with contextlib.redirect_stdout(io.StringIO()):
        genotyper.run_genotyping_overview(None, str(tmp_path), **SETTINGS)
    assert (tmp_path / 'predictions' / 'alleles.csv').read_text() == case['alleles_csv']


@pytest.mark.parametrize('case', GOLD['complex'], ids=[c['name'] for c in GOLD['complex']])
def test_complex_locus_genotype_matches_upstream(case, tmp_path):
    """run_genotyping_complex(locus_path, df): complex_alleles.csv and the printed summary byte for byte; a locus with one
    unit is left alone; where upstream's label lookup fails (one allele, and the row it asks for was set aside) so does this."""
    df = pd.DataFrame.from_dict(case['table'])
    buf = io.StringIO()
    np.random.seed(case['seed'])
    out_file = tmp_path / 'predictions' / 'complexSTR_analysis' / 'complex_alleles.csv'
    if case.get('error'):
        with pytest.raises(KeyError), contextlib.redirect_stdout(buf):
            genotyper.run_genotyping_complex(str(tmp_path), df, **SETTINGS)
        assert case['error'] == 'KeyError' and not out_file.exists()
        return
    with contextlib.redirect_stdout(buf):
        call = genotyper.run_genotyping_complex(str(tmp_path), df, **SETTINGS)
    if case['complex_alleles_csv'] is None:
        assert call is None and not out_file.exists()
    else:
        assert out_file.read_text() == case['complex_alleles_csv']
    assert buf.getvalue() == case['stdout']
    assert list(df.columns) == list(case['table'])  # the caller's table is not touched


def test_complex_table_is_read_from_disk_when_not_given(tmp_path):
    case = next(c for c in GOLD['complex'] if c['name'] == 'two_alleles_three_units')
    folder = tmp_path / 'predictions' / 'complexSTR_analysis'
    folder.mkdir(parents=True)
    assert genotyper.run_genotyping_complex(str(tmp_path)) is None  # no table: nothing to do
    pd.DataFrame.from_dict(case['table']).to_csv(folder / 'complex_repeat_units.csv')
    np.random.seed(case['seed'])
    with contextlib.redirect_stdout(io.StringIO()):
        call = genotyper.run_genotyping_complex(str(tmp_path), None, **SETTINGS)
    assert call.heterozygous and (folder / 'complex_alleles.csv').read_text() == case['complex_alleles_csv']
