"""Step 4 on the host: allele lengths from the per-read calls (consumer of the all-gathered results).

Mirrors src/genotyper/genotyping.py:60-66,183-214 and src/schemas/genotype.py: outlier filter (mean +- std_filter*std),
2-component tied Bayesian Gaussian mixture (scikit-learn), homozygous when a component's weight < min_weight,
allele = the called length nearest to the group's median.  O(reads) scalar work: it stays on the CPU.
"""
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np


def find_nearest(array: Sequence[int], value: float):
    array = np.asarray(array)
    return array[(np.abs(array - value)).argmin()]


@dataclass
class Genotype:
    group1: List[int]
    group2: List[int] = field(default_factory=list)
    predictions: List[int] = field(default_factory=list)

    @property
    def is_hetero(self) -> bool:
        return len(self.group2) > 0

    @property
    def first_allele(self):
        return int(find_nearest(self.group1, np.median(self.group1)))

    @property
    def second_allele(self):
        return int(find_nearest(self.group2, np.median(self.group2))) if self.is_hetero else '-'

    @property
    def first_allele_sz(self):
        return len(self.group1)

    @property
    def second_allele_sz(self):
        return len(self.group2) if self.is_hetero else '-'

    @property
    def alleles(self):
        return (self.first_allele, self.second_allele)


def filter_out(values: Sequence[int], std_coeff: float) -> List[int]:
    if len(values) <= 5:
        return list(values)
    mean, std = np.mean(values), np.std(values)
    return [i for i in values if (mean - std_coeff * std) <= i <= (mean + std_coeff * std)]


def run_bayes(X: np.ndarray, random_state=None):
    from sklearn.mixture import BayesianGaussianMixture
    return BayesianGaussianMixture(weight_concentration_prior=0.25, covariance_type='tied', n_components=2, n_init=5,
                                   max_iter=1000, random_state=random_state).fit(X)


def run_genotyping(unfilt_vals: Sequence[int], min_weight: float = 0.2, std_filter: float = 2, random_state=None) -> Genotype:
    vals = filter_out(unfilt_vals, std_filter)
    if len(np.unique(vals)) == 1:
        return Genotype(group1=vals)
    X = np.array(vals).reshape(-1, 1)
    model = run_bayes(X, random_state)
    if any(w < min_weight for w in model.weights_):
        return Genotype(group1=vals)
    preds = model.predict(X)
    return Genotype(group1=[i for i, g in zip(vals, preds) if g == 0], group2=[i for i, g in zip(vals, preds) if g == 1],
                    predictions=list(preds))


def genotype_results(results: np.ndarray, **kw) -> Genotype:
    """Genotype from the gathered wsx_result records (status == 0 reads only; `len2` is the allele length)."""
    ok = results['status'] == 0
    return run_genotyping([int(v) for v in results['len2'][ok]], **kw)


def store_predictions(gt: Genotype, locus_path: str, gt_bc: Optional[Genotype] = None) -> str:
    out = os.path.join(locus_path, 'predictions')
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, 'alleles.csv')
    with open(path, 'w') as f:
        f.write('WarpSTR_allele1,WarpSTR_allele1_freq,WarpSTR_allele2,WarpSTR_allele2_freq,'
                'basecall_allele1,basecall_allele1_freq,basecall_allele2,basecall_allele2_freq\n')
        f.write(f'{gt.first_allele},{gt.first_allele_sz},{gt.second_allele},{gt.second_allele_sz},')
        if gt_bc:
            f.write(f'{gt_bc.first_allele},{gt_bc.first_allele_sz},{gt_bc.second_allele},{gt_bc.second_allele_sz}')
    return path


def run_genotyping_overview(locus_path: str, overview=None, **kw) -> Genotype:
    """Genotype a locus from its overview.csv (`results` of the `saved` reads) and write predictions/alleles.csv
    (src/genotyper/genotyping.py:68-82,95-118)."""
    if overview is None:
        from .overview import load_overview
        _, overview = load_overview(locus_path)
    vals = [int(r.results) for r in overview.itertuples() if r.saved]
    gt = run_genotyping(vals, **kw)
    store_predictions(gt, locus_path)
    print(f'Allele lengths as given by WarpSTR: {gt.alleles}')
    return gt
