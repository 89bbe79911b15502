"""Complex-repeat post-processing on the host (strings; per locus / per called sequence).

Mirrors src/caller/wrapper.py:162-248:
  break_into_units(sequence)    -> (units, repeat_units, offsets)
  collapse_repeats(seq, ...)    -> per-unit counts of a called sequence
"""
from typing import List, Tuple

from .automata import IUPAC


def _expand_iupac(rep: str) -> List[str]:
    """All concrete strings of a pattern piece, first IUPAC alternative varying slowest per position
    in the order upstream produces them (src/caller/wrapper.py:201-215)."""
    out = ['']
    for ch in rep:
        if ch in IUPAC:
            out = [p + alt for alt in IUPAC[ch] for p in out]
        else:
            out = [p + ch for p in out]
    return out


def break_into_units(template: str) -> Tuple[List[str], List[List[str]], List[int]]:
    """Top-level bracketed units of a locus pattern, the concrete repeat strings of each, and the number of
    plain bases preceding each unit."""
    stack: List[int] = []
    units: List[str] = []
    offsets: List[int] = []
    offset = 0
    for idx, ch in enumerate(template):
        if ch in '({':
            stack.append(idx)
        elif ch in ')}':
            start = stack.pop()
            if not stack:
                units.append(template[start:idx + 1])
                offsets.append(offset)
                offset = 0
        elif not stack:
            offset += 1

    repeat_units: List[List[str]] = []
    for unit in units:
        pieces: List[str] = []
        cur = ''
        for ch in unit:
            if ch in '()':
                continue
            if ch == '{':
                pieces.append(cur)
                cur = ''
            elif ch == '}':
                pieces.extend([p + cur for p in pieces])
                cur = ''
            else:
                cur += ch
        if cur:
            pieces.append(cur)
        expanded: List[str] = []
        for p in pieces:
            expanded.extend(_expand_iupac(p))
        repeat_units.append(expanded)
    return units, repeat_units, offsets


def collapse_repeats(seq: str, repeat_units: List[List[str]], offsets: List[int], max_iter: int = 1_000_000):
    """Counts of every repeat string of every unit along a called sequence (greedy left-to-right scan with
    upstream's matching rule: when several alternatives match, each is counted and the LAST one advances)."""
    results = [[0] * len(u) for u in repeat_units]
    slide = seq
    for idx, (alts, off) in enumerate(zip(repeat_units, offsets)):
        slide = slide[off:]
        it = 0
        while slide:
            it += 1
            if it > max_iter:
                raise RuntimeError('collapse_repeats does not terminate for this pattern (empty repeat unit)')
            rest = None
            for k, alt in enumerate(alts):
                if alt == slide[:len(alt)]:
                    results[idx][k] += 1
                    rest = slide[len(alt):]
            if rest is None:
                break
            slide = rest
    return results
