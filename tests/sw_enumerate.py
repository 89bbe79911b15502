"""An independent Smith-Waterman for the tests of the flank localisation: the full score matrix with NumPy,
EVERY best end cell and EVERY co-optimal traceback, and find_sequence's arithmetic (src/extractor/tr_extractor.py:196-250)
carried out on aligned strings built the way Bio.pairwise2 pads them.  Shares no code with oracle/flank_oracle.c or the HIP
kernels, and has no tie rule of its own: where it finds exactly one optimal alignment, any correct local aligner -- pairwise2
included -- must return that one, so a hit that equals it is pinned whatever the tie rules are."""
import numpy as np


def sw_matrix(text: str, pat: str, match=2, mismatch=-3, gap=-3) -> np.ndarray:
    """H[i][j] = max(0, H[i-1][j-1] + s, H[i-1][j] + gap, H[i][j-1] + gap); rows are filled with a running maximum for the
    left-to-right chain (exact in integers: H[i][j] = max_k<=j (c[k] + gap (j - k)), c = the cell without its left neighbour)."""
    t = np.frombuffer(text.encode(), np.uint8)
    q = np.frombuffer(pat.encode(), np.uint8)
    n, p = len(t), len(q)
    H = np.zeros((n + 1, p + 1), np.int64)
    ramp = gap * np.arange(p + 1, dtype=np.int64)
    for i in range(1, n + 1):
        sub = np.where(q == t[i - 1], match, mismatch)
        c = np.maximum(0, np.maximum(H[i - 1, :-1] + sub, H[i - 1, 1:] + gap))
        c = np.concatenate(([0], c))
        H[i] = np.maximum.accumulate(c - ramp) + ramp
    return H


def optimal_alignments(text: str, pat: str, match=2, mismatch=-3, gap=-3, limit=64):
    """-> (best score, [end cells], [(i0, j0, i1, j1, ops)]) with every co-optimal traceback of every best end cell (at most
    `limit` alignments are spelled out).  ops: 'M' base against base, 'U' text base against a gap, 'L' pattern base against a gap."""
    H = sw_matrix(text, pat, match, mismatch, gap)
    best = int(H.max())
    if best <= 0:
        return 0, [], []
    ends = [(int(i), int(j)) for i, j in zip(*np.nonzero(H == best))]
    found = []
    for (bi, bj) in ends:
        stack = [(bi, bj, '')]
        while stack and len(found) < limit:
            i, j, ops = stack.pop()
            h = int(H[i, j])
            if h == 0 or i == 0 or j == 0:
                found.append((i, j, bi, bj, ops[::-1]))
                continue
            s = match if text[i - 1] == pat[j - 1] else mismatch
            if h == H[i - 1, j - 1] + s:
                stack.append((i - 1, j - 1, ops + 'M'))
            if h == H[i - 1, j] + gap:
                stack.append((i - 1, j, ops + 'U'))
            if h == H[i, j - 1] + gap:
                stack.append((i, j - 1, ops + 'L'))
    return best, ends, found


def hit_from_alignment(text: str, pat: str, i0: int, j0: int, i1: int, j1: int, ops: str, raw_score: int, gap_extend=-3):
    """find_sequence (tr_extractor.py:213-250) on pairwise2's output for this local alignment: full-length aligned strings
    (the shorter unaligned end padded with '-'), start/end = the local region inside them."""
    ti, pj, s1, s2 = i0, j0, [], []
    for o in ops:
        if o == 'M':
            s1.append(text[ti]); s2.append(pat[pj]); ti += 1; pj += 1
        elif o == 'U':
            s1.append(text[ti]); s2.append('-'); ti += 1
        else:
            s1.append('-'); s2.append(pat[pj]); pj += 1
    assert (ti, pj) == (i1, j1)
    al1 = '-' * max(j0 - i0, 0) + text[:i0] + ''.join(s1) + text[i1:]
    al2 = '-' * max(i0 - j0, 0) + pat[:j0] + ''.join(s2) + pat[j1:]
    al1 += '-' * (len(al2) - len(al1))
    al2 += '-' * (len(al1) - len(al2))
    start = max(i0, j0)
    end = start + len(ops)
    nums_gaps = al1[start:end].count('-')
    nums_gaps2 = al2[start:end].count('-')
    real_start = al2[:start].count('-')
    end = real_start + len(pat) + nums_gaps2 - nums_gaps
    ref, query = al1[real_start:end], al2[real_start:end]
    identity = sum(1 for x, y in zip(ref, query) if x == y)
    score = int(raw_score + (len(pat) - (len(query) - nums_gaps2)) * gap_extend)
    return dict(score=score, start=real_start, end=end, matches=identity, span=len(ref), row0=i0, col0=j0, row1=i1, col1=j1,
                gaps_text=nums_gaps, gaps_pattern=nums_gaps2, raw_score=raw_score, n_ops=len(ops))
