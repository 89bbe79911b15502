"""Step 4 on the host: from per-read allele lengths (what the GPUs all-gather) to the genotype of a locus.

Behaviour follows upstream's step 4 (src/genotyper/genotyping.py, src/schemas/genotype.py) and is pinned by fixtures recorded
from it (tests/golden/genotype.json); the code is this repository's own.  What upstream does, in words:

* simple repeats (genotyping.py:183-214): with more than five reads, reads further than `std_filter` population standard
  deviations from the mean are set aside; one distinct value left means one allele; otherwise a two-component Bayesian
  Gaussian mixture with a shared ("tied") covariance is fitted (scikit-learn, weight concentration prior 0.25, best of five
  starts, at most 1000 iterations) -- a component lighter than `min_weight` means one allele, else the reads split by their
  component.  An allele is reported as the member of its cluster nearest the cluster's median, with the cluster size
  (schemas/genotype.py:7-38);
* basecalled lengths `r_seq_start - l_seq_end` of the same reads are genotyped the same way when the overview has those
  columns (genotyping.py:95-103), and both go to predictions/alleles.csv (106-118);
* complex loci (genotyping.py:17-57, 121-180): the per-read table of repeat-unit counts is genotyped jointly over its unit
  columns; rows with any unit outside mean +- std_filter * std are set aside first, and the result goes to
  predictions/complexSTR_analysis/complex_alleles.csv.

scikit-learn draws the mixture's starting points from numpy's global generator unless `random_state` is given; the fixtures
seed it.  O(reads) scalar work per locus: it stays on the CPU.
"""
import os
from typing import Optional, Sequence, Tuple

import numpy as np

from .overview import COMPLEX_SUBDIR, PREDICTIONS_SUBDIR

NO_ALLELE = '-'
MIN_READS_FOR_TRIMMING = 6
# predictions/alleles.csv: allele and supporting reads, twice for the signal-based calls, twice for the basecalled lengths
ALLELE_COLUMNS = tuple(f'{source}_allele{k}{what}' for source in ('WarpSTR', 'basecall') for k in (1, 2) for what in ('', '_freq'))


def nearest_member(members, target: float):
    """The element of `members` closest to `target`; the earliest one when several are equally close."""
    arr = np.asarray(members)
    return arr[int(np.argmin(np.abs(arr - target)))]


class AlleleCall:
    """Reads of a locus split into one or two alleles.  `clusters[k]` holds the values assigned to allele k, in read order;
    `labels` the mixture component of every read that took part (empty when no mixture decided)."""

    def __init__(self, clusters: Sequence[Sequence[int]], labels: Sequence[int] = ()):
        first, rest = list(clusters[0]), [list(c) for c in clusters[1:2] if len(c)]
        self.clusters = tuple([first] + rest)
        self.labels = list(labels)

    @property
    def heterozygous(self) -> bool:
        return len(self.clusters) == 2 and len(self.clusters[1]) > 0

    def allele(self, k: int):
        if k == 1 and not self.heterozygous:
            return NO_ALLELE
        members = self.clusters[k]
        return int(nearest_member(members, np.median(members)))

    def support(self, k: int):
        if k == 1 and not self.heterozygous:
            return NO_ALLELE
        return len(self.clusters[k])

    @property
    def alleles(self) -> Tuple:
        return (self.allele(0), self.allele(1))

    def csv_fields(self) -> str:
        return ','.join(str(x) for k in (0, 1) for x in (self.allele(k), self.support(k)))

    def __repr__(self):
        return f'AlleleCall(alleles={self.alleles}, support=({self.support(0)}, {self.support(1)}))'


def trim_outliers(values: Sequence[int], n_std: float) -> list:
    """Values within n_std population standard deviations of the mean (bounds included); small sets are left alone."""
    values = list(values)
    if len(values) < MIN_READS_FOR_TRIMMING:
        return values
    centre, spread = np.mean(values), np.std(values)
    low, high = centre - n_std * spread, centre + n_std * spread
    return [v for v in values if not (v < low or v > high)]


def fit_allele_mixture(points: np.ndarray, random_state=None):
    """Two Gaussians with one shared covariance; few reads should not make a second allele: a sparse Dirichlet prior."""
    from sklearn.mixture import BayesianGaussianMixture
    mixture = BayesianGaussianMixture(n_components=2, covariance_type='tied', weight_concentration_prior=0.25, n_init=5,
                                      max_iter=1000, random_state=random_state)
    return mixture.fit(points)


def _is_single_allele(mixture, min_weight: float) -> bool:
    return bool(np.min(mixture.weights_) < min_weight)


def call_alleles(lengths: Sequence[int], min_weight: float = 0.2, std_filter: float = 2, random_state=None) -> AlleleCall:
    """Genotype one list of per-read lengths."""
    kept = trim_outliers(lengths, std_filter)
    if len(np.unique(kept)) == 1:
        return AlleleCall([kept])
    column = np.array(kept).reshape(-1, 1)
    mixture = fit_allele_mixture(column, random_state)
    if _is_single_allele(mixture, min_weight):
        return AlleleCall([kept])
    labels = mixture.predict(column)
    split = ([v for v, lab in zip(kept, labels) if lab == 0], [v for v, lab in zip(kept, labels) if lab == 1])
    return AlleleCall(split, labels)


def genotype_results(results: np.ndarray, **settings) -> AlleleCall:
    """From the gathered wsx_result records: reads with status 0, allele length = `len2` (the second pass's call)."""
    called = results['status'] == 0
    return call_alleles([int(v) for v in results['len2'][called]], **settings)


def write_alleles_csv(locus_path: str, warpstr: AlleleCall, basecall: Optional[AlleleCall] = None) -> str:
    """predictions/alleles.csv: a header and one record; the basecall half stays empty without basecalled lengths, and the
    record carries no line end (as upstream writes it)."""
    folder = os.path.join(locus_path, PREDICTIONS_SUBDIR)
    os.makedirs(folder, exist_ok=True)
    path = os.path.join(folder, 'alleles.csv')
    record = warpstr.csv_fields() + ',' + (basecall.csv_fields() if basecall is not None else '')
    with open(path, 'w') as out:
        out.write(','.join(ALLELE_COLUMNS) + '\n' + record)
    return path


def lengths_from_overview(overview):
    """(signal-based lengths, basecalled lengths or None) of the reads step 3 called (`saved` rows, overview order)."""
    saved = overview[np.asarray(overview['saved']).astype(bool)]
    called = [v for v in saved['results']]
    if 'r_seq_start' in overview.columns and 'l_seq_end' in overview.columns:
        return called, [r - l for r, l in zip(saved['r_seq_start'], saved['l_seq_end'])]
    return called, None


def run_genotyping_overview(overview, locus_path: str, muscle_path: Optional[str] = None, **settings) -> AlleleCall:
    """Upstream's entry point (WarpSTR.py:78; genotyping.py:68-82), same argument order: genotype the locus from its overview
    table (read from <locus_path>/overview.csv when None) and write predictions/alleles.csv.  Plots and the MUSCLE
    alignment of the called sequences are outside this repository's scope (`muscle_path` is accepted and ignored)."""
    if overview is None:
        from .overview import load_overview
        _, overview = load_overview(locus_path)
    called, basecalled = lengths_from_overview(overview)
    warpstr = call_alleles(called, **settings)
    basecall = call_alleles(basecalled, **settings) if basecalled is not None else None
    write_alleles_csv(locus_path, warpstr, basecall)
    print(f'Allele lengths as given by WarpSTR: {warpstr.alleles}')
    if basecall is not None:
        print(f'Allele lengths as given by basecall: {basecall.alleles}')
    return warpstr


# ---------------------------------------------------------------------------------------------------------------------
# complex loci: several repeat units per read
# ---------------------------------------------------------------------------------------------------------------------
class ComplexCall:
    """Joint genotype over the repeat units of a complex locus: per unit the repeat count of each allele."""

    def __init__(self, units, first, first_support, second=None, second_support=None):
        self.units = list(units)
        self.first, self.first_support = list(first), first_support
        self.second, self.second_support = (list(second) if second is not None else None), second_support

    @property
    def heterozygous(self) -> bool:
        return self.second is not None

    def rows(self):
        for i, unit in enumerate(self.units):
            yield unit, self.first[i], (self.second[i] if self.heterozygous else NO_ALLELE)


def _rows_outside(table, units, n_std: float) -> list:
    """Row numbers with any unit count strictly outside mean +- n_std * std of its column."""
    flagged = set()
    for unit in units:
        column = table[unit]
        centre, spread = np.mean(column), np.std(column)
        low, high = centre - n_std * spread, centre + n_std * spread
        flagged.update(row for row, count in enumerate(column) if count < low or count > high)
    return sorted(flagged)


def call_complex_alleles(table, min_weight: float = 0.2, std_filter: float = 2, random_state=None) -> Optional[ComplexCall]:
    """Genotype a per-read table of repeat-unit counts (columns = units, plus `reverse`).  None when the locus has fewer
    than two units.  Two upstream habits are kept because they decide the result: the rows set aside are addressed by ROW
    NUMBER in an index of labels (identical for the 0..n-1 index the table is written and read with), and the one-allele
    case looks its representative up by LABEL = row number of the remaining rows -- a KeyError if that row was set aside
    (genotyping.py:36-40, schemas/genotype.py:37-38; recorded in tests/golden/genotype.json)."""
    units = [name for name in table.columns if name != 'reverse']
    if len(units) < 2:
        return None
    kept = table
    if len(table) >= MIN_READS_FOR_TRIMMING:
        kept = table.drop(index=_rows_outside(table, units, std_filter))
    points = np.array(kept[units]).reshape(-1, len(units))
    mixture = fit_allele_mixture(points, random_state)
    if _is_single_allele(mixture, min_weight):
        representative = []
        for unit in units:
            column = kept[unit]
            row = int(np.argmin(np.abs(np.asarray(column) - np.median(column))))
            representative.append(column.loc[row])
        return ComplexCall(units, representative, len(kept))
    labels = mixture.predict(points)
    halves = []
    for component in (0, 1):
        member_rows = kept[units][labels == component]
        halves.append(([int(nearest_member(member_rows[unit].values, np.median(member_rows[unit]))) for unit in units],
                       int(np.sum(labels == component))))
    return ComplexCall(units, halves[0][0], halves[0][1], halves[1][0], halves[1][1])


def run_genotyping_complex(locus_path: str, df=None, **settings) -> Optional[ComplexCall]:
    """Upstream's entry point for complex loci (WarpSTR.py:79; genotyping.py:121-180): `df` is the table main_wrapper
    returned (or None: read predictions/complexSTR_analysis/complex_repeat_units.csv if it exists).  Writes
    complex_alleles.csv next to it and prints upstream's summary; the plot is outside this repository's scope."""
    folder = os.path.join(locus_path, PREDICTIONS_SUBDIR, COMPLEX_SUBDIR)
    if df is None:
        source = os.path.join(folder, 'complex_repeat_units.csv')
        if not os.path.isfile(source):
            return None
        import pandas as pd
        df = pd.read_csv(source, index_col=0)
    call = call_complex_alleles(df, **settings)
    if call is None:
        return None
    if call.heterozygous:
        print('Genotyped complex repeats in 2 alleles:')
        header = 'unit,allele1_repeats,allele2_repeats'
    else:
        print('Genotyped complex repeats in a homozygous allele:')
        header = 'unit,allele1_repeats, allele2_repeats'  # (upstream's header of this case has the blank)
    for unit, a1, a2 in call.rows():
        print(f'Unit: {unit:10} Repeats: {a1:5} {a2:5}')
    print(f'There were {call.first_support} reads for allele1 and '
          f'{call.second_support if call.heterozygous else NO_ALLELE} for allele2')
    os.makedirs(folder, exist_ok=True)
    with open(os.path.join(folder, 'complex_alleles.csv'), 'w') as out:
        out.write(header + '\n')
        for unit, a1, a2 in call.rows():
            out.write(f'{unit},{a1},{a2}\n')
    return call
