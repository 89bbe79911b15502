"""Step-3 outputs in the reference's on-disk formats (src/caller/overview.py:37-115,
src/extractor/tr_extractor.py:108-140): overview.csv columns, FASTA files, complex-unit CSV."""
import os
from typing import List, Sequence, Tuple

import numpy as np
import pandas as pd

OVERVIEW_NAME = 'overview.csv'
PREDICTIONS_SUBDIR = 'predictions'
COMPLEX_SUBDIR = 'complexSTR_analysis'
LOCUS_INFO_SUBDIR = 'expected_signals'
LOCUS_FLANKS = 'sequences.csv'


def load_overview(locus_path: str):
    overview_path = os.path.join(locus_path, OVERVIEW_NAME)
    try:
        df = pd.read_csv(overview_path)
        df.set_index('read_name', inplace=True)
        df.columns = df.columns.map(str)
    except FileNotFoundError:
        raise FileNotFoundError(f'Not found the overview file {overview_path} - Please check the "output" in config')
    return overview_path, df


def table_from_text(text: str):
    """The DataFrame load_overview would give for a file with this content."""
    import io
    df = pd.read_csv(io.StringIO(text))
    df.set_index('read_name', inplace=True)
    df.columns = df.columns.map(str)
    return df


def load_flanks(locus_path: str) -> Tuple[str, str, str, str]:
    """(left_template, right_template, left_reverse, right_reverse) from expected_signals/sequences.csv."""
    path = os.path.join(locus_path, LOCUS_INFO_SUBDIR, LOCUS_FLANKS)
    if not os.path.exists(path):
        raise FileNotFoundError(f'File with flanks not found in path={path}')
    with open(path, 'r') as f:
        cols = [c.rstrip() for c in f.readline().split(',')]
        seqid = cols.index('sequence') if 'sequence' in cols else 0
        rows = [f.readline().split(',')[seqid].rstrip() for _ in range(4)]
    return rows[0], rows[1], rows[2], rows[3]


FLANK_NAMES = ['left_flank_template', 'right_flank_template', 'left_flank_reverse', 'right_flank_reverse']


def store_flanks(locus_path: str, flanks: Sequence[str]):
    """Write the flank file load_flanks reads, in the squiggler step's format (src/squiggler/Squiggler.py:69-75); for
    callers that get the flanks from somewhere other than a reference genome."""
    os.makedirs(os.path.join(locus_path, LOCUS_INFO_SUBDIR), exist_ok=True)
    with open(os.path.join(locus_path, LOCUS_INFO_SUBDIR, LOCUS_FLANKS), 'w') as f:
        f.write('type,sequence\n')
        for name, seq in zip(FLANK_NAMES, flanks):
            f.write(f'{name},{seq.upper()}\n')


def result_columns(df_overview, seq_results: Sequence[Tuple[str, str]], cost_results: Sequence[Tuple[float, float]]):
    """Per-row values of the four step-3 columns: called reads in overview order, -1 for rows that were not `saved`
    (src/caller/overview.py:57-73).  Also returns the (read, resc_seq, reverse) triples for the FASTA files."""
    saved = np.asarray(df_overview['saved']).astype(bool)
    n_saved = int(saved.sum())
    if n_saved != len(seq_results) or n_saved != len(cost_results):
        raise ValueError(f'{n_saved} saved reads in the overview but {len(seq_results)} results')
    cols = {name: np.full(len(saved), -1, dtype=dt) for name, dt in
            (('results', np.int64), ('orig', np.int64), ('dtw_cost1', np.float64), ('dtw_cost2', np.float64))}
    where = np.flatnonzero(saved)
    cols['results'][where] = [len(s[1]) for s in seq_results]
    cols['orig'][where] = [len(s[0]) for s in seq_results]
    cols['dtw_cost1'][where] = [c[0] for c in cost_results]
    cols['dtw_cost2'][where] = [c[1] for c in cost_results]
    names = df_overview.index.to_numpy()[where]
    strands = np.asarray(df_overview['reverse']).astype(bool)[where]
    fasta = [(str(nm), s[1], bool(rv)) for nm, s, rv in zip(names, seq_results, strands)]
    return cols, fasta


def write_results_to_fasta(fasta, locus_path: str):
    """predictions/sequences/{all,sequences_template,sequences_reverse}.fasta; record = '>id', sequence, blank line
    (src/caller/overview.py:76-100)."""
    base = os.path.join(locus_path, PREDICTIONS_SUBDIR, 'sequences')
    os.makedirs(base, exist_ok=True)
    selections = (('all.fasta', None), ('sequences_template.fasta', False), ('sequences_reverse.fasta', True))
    for fname, strand in selections:
        with open(os.path.join(base, fname), 'w') as f:
            f.writelines(f'>{rid}\n{seq}\n\n' for rid, seq, rev in fasta if strand is None or rev == strand)


def store_results(overview_path, df_overview, seq_results, cost_results, locus_path: str, write: bool = True):
    """Write the FASTA files and the overview with (re)placed `results, orig, dtw_cost1, dtw_cost2` columns; any older
    column starting with 'result' is dropped first (src/caller/overview.py:48-54,103-115)."""
    cols, fasta = result_columns(df_overview, seq_results, cost_results)
    if write:
        write_results_to_fasta(fasta, locus_path)
    stale = [c for c in df_overview.columns if c.startswith('result')]
    df_overview = df_overview.drop(columns=stale)
    for name in ('results', 'orig', 'dtw_cost1', 'dtw_cost2'):
        df_overview[name] = cols[name]
    if write:
        df_overview.to_csv(overview_path)
    return df_overview


def store_collapsed(results, units: List[str], rep_units: List[List[str]], reverse_lst: List[bool], locus_path: str,
                    write: bool = True):
    """predictions/complexSTR_analysis/complex_repeat_units.csv (src/caller/overview.py:11-34): one row per read;
    a unit with alternatives gives `main_<first>` (all its counts summed) plus one `inter_<suffix>` column per further
    alternative, a plain unit gives one column named after its bases; last column `reverse`."""
    table = {}
    for u, (unit, alts) in enumerate(zip(units, rep_units)):
        counts = np.array([r[u] for r in results], dtype=np.int64).reshape(len(results), len(alts))   # (no reads: an empty table; upstream's IndexError)
        if counts.shape[1] > 1:
            table['main_' + alts[0]] = counts.sum(axis=1)
            for a_idx in range(1, len(alts)):
                table['inter_' + alts[a_idx][len(alts[0]):]] = counts[:, a_idx]
        else:
            table[unit.strip('(').strip(')')] = counts[:, 0]
    table['reverse'] = list(reverse_lst)
    df = pd.DataFrame(table)
    if write:
        out = os.path.join(locus_path, PREDICTIONS_SUBDIR, COMPLEX_SUBDIR)
        os.makedirs(out, exist_ok=True)
        df.to_csv(os.path.join(out, 'complex_repeat_units.csv'))
    return df
