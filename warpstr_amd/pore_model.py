"""Expected k-mer current levels (the squiggler side of the hot path).

Mirrors the behaviour of the reference's ``PoreModel`` (src/squiggler/pore_model.py:13-47):
the ONT 6-mer table's ``level_mean`` column is MAD-normalised over all 4096 k-mers
(src/schemas/fast5.py:104-114) and looked up by k-mer.  Here the table is a flat f64 array
indexed by the base-4 code of the k-mer (A=0,C=1,G=2,T=3, first base most significant), which
is also the layout the HIP library receives (``state value`` per automaton state).

Data provenance: ``data/r94_6mer_level_mean.npy`` holds the 4096 ``level_mean`` values of the
ONT r9.4 450 bps 6-mer template model shipped upstream as
``example/deps/template_median68pA.model`` (rows are in lexicographic k-mer order).  A TSV in that
same format can be given instead (config key ``pore_model_path``).
"""
import os
from typing import Optional

import numpy as np

_BASE_CODE = {'A': 0, 'C': 1, 'G': 2, 'T': 3}
_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'r94_6mer_level_mean.npy')


def normalize_signal_mad(data):
    """Median/MAD normalisation; same arithmetic as src/schemas/fast5.py:104-114."""
    data = np.asarray(data)
    shift = np.mean(np.percentile(data, (46.5, 53.5)))
    scale = np.median(np.abs(data - shift))
    return np.asarray((data - shift) / scale)


def kmer_code(kmer: str) -> int:
    code = 0
    for ch in kmer:
        code = code * 4 + _BASE_CODE[ch]
    return code


class PoreModel:
    """k-mer -> normalised expected level."""

    def __init__(self, pore_model_path: Optional[str] = None):
        if pore_model_path is None:
            level_mean = np.load(_DATA)
            self.kmersize = 6
        else:
            level_mean, self.kmersize = self._read_tsv(pore_model_path)
        if level_mean.shape[0] != 4 ** self.kmersize:
            raise ValueError('pore model table must hold every k-mer exactly once')
        self.level_norm = normalize_signal_mad(level_mean)

    @staticmethod
    def _read_tsv(path: str):
        """`level_mean` by k-mer from a tab-separated model table (src/squiggler/pore_model.py:15-33).  The numbers are parsed
        by pandas' reader, as upstream's are: its default float converter is not correctly rounded, a table written with 17
        significant digits comes out a last bit different from float() -- and the state levels with it."""
        if not os.path.exists(path):
            raise FileNotFoundError('Not found pore model table at path', path)
        from pandas import read_csv
        table = read_csv(path, sep='\t', header=0)
        if 'kmer' not in table or 'level_mean' not in table:
            raise ValueError('Pore model table do not contains "kmer" and "level_mean" columns')
        kmers = [str(k) for k in table['kmer']]
        k = len(kmers[0])
        out = np.full(4 ** k, np.nan)
        seen = np.zeros(4 ** k, bool)
        for kmer, level in zip(kmers, np.asarray(table['level_mean'], dtype=np.float64)):
            code = kmer_code(kmer)
            if not seen[code]:   # (upstream's look-up takes the first row of a k-mer)
                out[code], seen[code] = level, True
        if not seen.all():
            raise ValueError('pore model table is missing k-mers')
        return out, k

    def get_value(self, kmer: str) -> float:
        """Level of one k-mer (src/squiggler/pore_model.py:45-47)."""
        return float(self.level_norm[kmer_code(kmer)])

    def levels_for(self, sequence: str) -> np.ndarray:
        """Expected level of every k-mer of a plain sequence (src/squiggler/Squiggler.py:20-28)."""
        k = self.kmersize
        return np.array([self.get_value(sequence[i:i + k]) for i in range(len(sequence) - k + 1)], dtype=np.float64)


    def _get_consecutive_diff(self, pattern: str):
        """Mean and median |difference| between the expected levels of consecutive k-mers of a repeated unit
        (src/squiggler/pore_model.py:34-43).  Like upstream, a one-base unit runs past the repeated pattern
        (IndexError there; here too).  np.mean / np.median of the handful of values are restated on Python floats (NumPy's
        pairwise sum: one accumulator below eight values, eight above; the median as the mean of the middle pair) -- the same
        doubles, a tenth of the time: this runs for every unit of every locus of a run (tests/test_host_logic.py holds the two
        forms against each other)."""
        k = self.kmersize
        rep = pattern * k
        if len(pattern) + k > len(rep):
            raise IndexError(f'repeat unit {pattern!r} is too short for {k}-mers')
        level = self.level_norm
        vals = [float(level[kmer_code(rep[i:k + i])]) for i in range(len(pattern) + 1)]
        diffs = [abs(b - a) for a, b in zip(vals, vals[1:])]
        n = len(diffs)
        if n == 0 or n > 128:
            arr = np.abs(np.diff(vals))
            return np.mean(arr), np.median(arr)
        ordered = sorted(diffs)
        med = ordered[n // 2] if n & 1 else (ordered[n // 2 - 1] + ordered[n // 2]) / 2.0
        return np.float64(_pairwise_sum(diffs) / n), np.float64(med)

    def get_diffs_for_all(self, sequence: str):
        """{unit: (mean_diff, median_diff)} for every bracketed unit of a locus pattern, IUPAC codes expanded
        (src/squiggler/pore_model.py:49-71; same insertion order, later duplicates overwrite earlier ones)."""
        diffs = {}
        for r in _UNIT_RE.findall(sequence):
            patterns = ['']
            for char in (c for c in r if c not in '(){}'):
                if char not in _IUPAC:
                    patterns = [p + char for p in patterns]
                else:
                    patterns = [p + alt for alt in _IUPAC[char] for p in patterns]
            for p in patterns:
                diffs[p] = self._get_consecutive_diff(p)
        return diffs


import re  # noqa: E402

_UNIT_RE = re.compile(r'[\(\{].*?[\)\}]')
_IUPAC = {'R': 'AG', 'Y': 'CT', 'S': 'GC', 'W': 'AT', 'K': 'GT', 'M': 'AC', 'B': 'CGT', 'D': 'AGT', 'H': 'ACT', 'V': 'ACG',
          'N': 'ACGT'}  # src/templates.py:32-44


def _pairwise_sum(a):
    """np.add.reduce of up to 128 doubles (numpy/core/src/umath/loops_utils.h.src: pairwise_sum)."""
    n = len(a)
    if n < 8:
        res = 0.0
        for x in a:
            res += x
        return res
    r = list(a[:8])
    i = 8
    while i < n - (n % 8):
        for j in range(8):
            r[j] += a[i + j]
        i += 8
    res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
    while i < n:
        res += a[i]
        i += 1
    return res


_default: Optional[PoreModel] = None


def default_pore_model() -> PoreModel:
    global _default
    if _default is None:
        _default = PoreModel()
    return _default
