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


def append_results(seq_results: Sequence[Tuple[str, str]], cost_results: Sequence[Tuple[float, float]], df_overview):
    fasta_lst, newcol, dbg1, dbg2, dbg3 = [], [], [], [], []
    idx = 0
    for row in df_overview.itertuples():
        if row.saved:
            newcol.append(len(seq_results[idx][1]))
            dbg1.append(len(seq_results[idx][0]))
            dbg2.append(cost_results[idx][0])
            dbg3.append(cost_results[idx][1])
            fasta_lst.append((row.Index, seq_results[idx][1], row.reverse))
            idx += 1
        else:
            newcol.append(-1)
            dbg1.append(-1)
            dbg2.append(-1)
            dbg3.append(-1)
    return fasta_lst, newcol, dbg1, dbg2, dbg3


def write_results_to_fasta(fasta_lst, locus_path: str):
    base = os.path.join(locus_path, PREDICTIONS_SUBDIR, 'sequences')
    os.makedirs(base, exist_ok=True)
    targets = {'all.fasta': lambda rev: True, 'sequences_template.fasta': lambda rev: rev is False or rev == 0,
               'sequences_reverse.fasta': lambda rev: bool(rev)}
    for name, keep in targets.items():
        with open(os.path.join(base, name), 'w') as f:
            for fid, seq, rev in fasta_lst:
                rev = bool(rev) if isinstance(rev, (np.bool_, bool, int, np.integer)) else rev
                if name == 'all.fasta' or keep(rev):
                    f.write('>' + fid + '\n' + seq + '\n\n')


def save_overview(overview_path, df_overview, newcol, dbg1, dbg2, dbg3):
    prev = [c for c in df_overview.columns if c.startswith('result')]
    df_overview.drop(columns=prev, inplace=True)
    df_overview['results'] = newcol
    df_overview['orig'] = dbg1
    df_overview['dtw_cost1'] = dbg2
    df_overview['dtw_cost2'] = dbg3
    df_overview.to_csv(overview_path)
    return df_overview


def store_results(overview_path, df_overview, seq_results, cost_results, locus_path: str):
    fasta_lst, newcol, dbg1, dbg2, dbg3 = append_results(seq_results, cost_results, df_overview)
    write_results_to_fasta(fasta_lst, locus_path)
    return save_overview(overview_path, df_overview, newcol, dbg1, dbg2, dbg3)


def store_collapsed(results, units: List[str], rep_units: List[List[str]], reverse_lst: List[bool], locus_path: str):
    preds = {}
    for idx, unit in enumerate(units):
        if len(results[0][idx]) > 1:
            preds['main_' + rep_units[idx][0]] = np.array([np.sum(j[idx]) for j in results])
            for idx2, k in enumerate(rep_units[idx][1:]):
                preds['inter_' + k[len(rep_units[idx][0]):]] = np.array([j[idx][idx2 + 1] for j in results])
        else:
            preds[unit.strip('(').strip(')')] = np.array([j[idx][0] for j in results])
    preds['reverse'] = reverse_lst
    df = pd.DataFrame.from_dict(preds)
    out = os.path.join(locus_path, PREDICTIONS_SUBDIR, COMPLEX_SUBDIR)
    os.makedirs(out, exist_ok=True)
    df.to_csv(os.path.join(out, 'complex_repeat_units.csv'))
    return df
