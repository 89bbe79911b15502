"""Seeded synthetic squiggles for tests and bench.

A read is: random flanks + one instantiation of the locus pattern (loops unrolled a random number
of times, optional blocks kept or dropped, IUPAC codes resolved) -> k-mer level sequence from the
pore model -> dwell of >= ``min_dwell`` samples per k-mer summing to exactly T -> Gaussian noise.
Everything is float64, as the reference's normalised squiggles are.
"""
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np

from .automata import IUPAC, AutomatonTable, compile_automaton, reverse_pattern
from .pore_model import PoreModel, default_pore_model


def random_flank(rng: np.random.Generator, length: int) -> str:
    return ''.join('ACGT'[i] for i in rng.integers(0, 4, size=length))


def instantiate_pattern(pattern: str, rng: np.random.Generator, lo: int, hi: int) -> Tuple[str, List[int]]:
    """One concrete base string matched by ``pattern`` and the unroll count of each top-level loop."""
    counts: List[int] = []

    def expand(sub: str, depth: int) -> str:
        out, i = '', 0
        while i < len(sub):
            ch = sub[i]
            if ch == '(':
                j = _match(sub, i, '(', ')')
                n = int(rng.integers(lo, hi + 1)) if depth == 0 else int(rng.integers(1, 3))
                if depth == 0:
                    counts.append(n)
                out += ''.join(expand(sub[i + 1:j], depth + 1) for _ in range(n))
                i = j + 1
            elif ch == '{':
                j = _match(sub, i, '{', '}')
                if rng.integers(0, 2):
                    out += expand(sub[i + 1:j], depth + 1)
                i = j + 1
            elif ch in IUPAC:
                alts = IUPAC[ch]
                out += alts[int(rng.integers(0, len(alts)))]
                i += 1
            else:
                out += ch
                i += 1
        return out

    return expand(pattern, 0), counts


def _match(s: str, i: int, op: str, cl: str) -> int:
    depth = 0
    for j in range(i, len(s)):
        if s[j] == op:
            depth += 1
        elif s[j] == cl:
            depth -= 1
            if depth == 0:
                return j
    raise ValueError('unbalanced pattern')


@dataclass
class SyntheticLocus:
    pattern: str
    flank_length: int
    left_t: str
    right_t: str
    template: AutomatonTable
    reverse: AutomatonTable

    @property
    def left_r(self) -> str:
        return _revcomp(self.right_t)

    @property
    def right_r(self) -> str:
        return _revcomp(self.left_t)


_COMP = str.maketrans('ACGT', 'TGCA')


def _revcomp(s: str) -> str:
    return s.translate(_COMP)[::-1]


def make_locus(pattern: str, flank_length: int, seed: int, pore_model: Optional[PoreModel] = None,
               max_states: Optional[int] = None, max_tries: int = 200) -> SyntheticLocus:
    """Random flanks for a locus; optionally re-draw until both automata have <= max_states states."""
    rng = np.random.default_rng(seed)
    for _ in range(max_tries):
        left, right = random_flank(rng, flank_length), random_flank(rng, flank_length)
        tmp = compile_automaton(left + pattern + right, pore_model)
        rev = compile_automaton(_revcomp(right) + reverse_pattern(pattern) + _revcomp(left), pore_model)
        if max_states is None or max(tmp.n_states, rev.n_states) <= max_states:
            return SyntheticLocus(pattern, flank_length, left, right, tmp, rev)
    raise RuntimeError(f'no flanks found giving <= {max_states} states for {pattern} at flank {flank_length}')


def squiggle(locus: SyntheticLocus, reverse: bool, T: int, rng: np.random.Generator, lo: int = 5, hi: int = 30,
             sigma: float = 0.25, min_dwell: int = 4, pore_model: Optional[PoreModel] = None
             ) -> Tuple[np.ndarray, List[int]]:
    """One synthetic normalised squiggle of exactly T samples and its planted loop counts."""
    pm = pore_model or default_pore_model()
    pattern = reverse_pattern(locus.pattern) if reverse else locus.pattern
    left, right = (locus.left_r, locus.right_r) if reverse else (locus.left_t, locus.right_t)
    for _ in range(64):
        body, counts = instantiate_pattern(pattern, rng, lo, hi)
        levels = pm.levels_for(left + body + right)
        if min_dwell * len(levels) <= T:
            break
    else:
        raise ValueError('T too short for this locus at min_dwell samples per k-mer')
    extra = T - min_dwell * len(levels)
    dwell = min_dwell + rng.multinomial(extra, np.full(len(levels), 1.0 / len(levels)))
    sig = np.repeat(levels, dwell) + rng.normal(0.0, sigma, size=T)
    return np.ascontiguousarray(sig, dtype=np.float64), counts


def batch(locus: SyntheticLocus, n_reads: int, T, seed: int, lo: int = 5, hi: int = 30, sigma: float = 0.25,
          reverse_fraction: float = 0.5):
    """n_reads squiggles (T fixed int, or (Tmin, Tmax) for ragged) -> (signals list, reverse flags, counts)."""
    rng = np.random.default_rng(seed)
    sigs, revs, truth = [], [], []
    for _ in range(n_reads):
        rev = bool(rng.random() < reverse_fraction)
        t = int(T) if np.isscalar(T) else int(rng.integers(T[0], T[1] + 1))
        s, c = squiggle(locus, rev, t, rng, lo, hi, sigma)
        sigs.append(s)
        revs.append(rev)
        truth.append(c)
    return sigs, revs, truth
