"""Flank localisation, CPU side: the C oracle (oracle/flank_oracle.c) against an independent, literal
pure-Python statement of the same rules, against hand-checked answers, and against properties of the reference's
find_sequence arithmetic (src/extractor/tr_extractor.py:196-250).  Parity with Biopython itself is unpinned (absent)."""
import numpy as np
import pytest

from oracle import flank


def py_find_sequence(text: str, pat: str, match=2, mismatch=-3, gap=-3):
    """Full-matrix Smith-Waterman + traceback with the documented tie rules, then find_sequence's string arithmetic
    performed on actually built aligned strings (as upstream does on pairwise2's output)."""
    n, p = len(text), len(pat)
    H = [[0] * (p + 1) for _ in range(n + 1)]
    best, bi, bj = 0, 0, 0
    for i in range(1, n + 1):
        for j in range(1, p + 1):
            h = max(0, H[i - 1][j - 1] + (match if text[i - 1] == pat[j - 1] else mismatch), H[i - 1][j] + gap, H[i][j - 1] + gap)
            H[i][j] = h
            if h > 0 and h >= best:
                best, bi, bj = h, i, j
    if best == 0:
        return None
    wr = p + (match * p) // (-gap) + 2
    i0 = max(bi - wr, 0)
    W = [[0] * (p + 1) for _ in range(bi - i0 + 1)]
    for i in range(i0 + 1, bi + 1):
        for j in range(1, p + 1):
            W[i - i0][j] = max(0, W[i - i0 - 1][j - 1] + (match if text[i - 1] == pat[j - 1] else mismatch), W[i - i0 - 1][j] + gap,
                               W[i - i0][j - 1] + gap)
    i, j, a1, a2 = bi, bj, [], []
    while i > i0 and j > 0 and W[i - i0][j] > 0:
        h = W[i - i0][j]
        if h == W[i - i0 - 1][j - 1] + (match if text[i - 1] == pat[j - 1] else mismatch):
            a1.append(text[i - 1]); a2.append(pat[j - 1]); i -= 1; j -= 1
        elif h == W[i - i0 - 1][j] + gap:
            a1.append(text[i - 1]); a2.append('-'); i -= 1
        else:
            a1.append('-'); a2.append(pat[j - 1]); j -= 1
    a1.reverse(); a2.reverse()
    # pairwise2-style full-length aligned strings (_finish_backtrace pads the shorter unaligned end with gaps)
    al1 = '-' * max(j - i, 0) + text[:i] + ''.join(a1) + text[bi:]
    al2 = '-' * max(i - j, 0) + pat[:j] + ''.join(a2) + pat[bj:]
    al1 += '-' * (len(al2) - len(al1))
    al2 += '-' * (len(al1) - len(al2))
    start, end = max(i, j), max(i, j) + len(a1)
    # tr_extractor.py:226-246, literally
    nums_gaps = al1[start:end].count('-')
    nums_gaps2 = al2[start:end].count('-')
    real_start = al2[:start].count('-')
    end = real_start + len(pat) + nums_gaps2 - nums_gaps
    ref, query = al1[real_start:end], al2[real_start:end]
    identity = sum(1 for x, y in zip(ref, query) if x == y)
    score = int(best + (len(pat) - (len(query) - nums_gaps2)) * gap)
    return dict(score=score, start=real_start, end=end, matches=identity, span=len(ref), row0=i, col0=j, row1=bi, col1=bj,
                gaps_text=nums_gaps, gaps_pattern=nums_gaps2, raw_score=best)


def mutate(rng, s, rate):
    out = []
    for ch in s:
        u = rng.random()
        if u < rate / 3:
            continue                                   # deletion
        if u < 2 * rate / 3:
            out.append('ACGT'[rng.integers(4)])        # substitution (may coincide)
            continue
        out.append(ch)
        if u > 1 - rate / 3:
            out.append('ACGT'[rng.integers(4)])        # insertion
    return ''.join(out)


def random_case(rng, n, p, rate, edge=None):
    text = ''.join('ACGT'[k] for k in rng.integers(0, 4, size=n))
    a = int(rng.integers(0, max(n - p, 1)))
    if edge == 'head':
        a = 0
    if edge == 'tail':
        a = max(n - p, 0)
    return text, mutate(rng, text[a:a + p], rate)


def test_known_answers():
    text = 'TTTTT' + 'ACGTACGGTCA' + 'GGGGG'
    h = flank.find_sequence(text.encode(), b'ACGTACGGTCA')
    assert (h.status, h.score, h.start, h.end, h.matches, h.span, h.n_ops) == (0, 22, 5, 16, 11, 11, 11)
    # one substitution in the middle: 10 matches, 1 mismatch -> 20 - 3 = 17; still the whole flank
    h = flank.find_sequence(text.encode(), b'ACGTAGGGTCA')
    assert (h.score, h.start, h.end, h.matches, h.span, h.ops) == (17, 5, 16, 10, 11, b'M' * 11)
    # the flank's first two bases are not in the read: the local alignment starts at pattern base 2, the position is
    # extrapolated to where the whole flank would start (real_start = row0 - col0)
    h = flank.find_sequence(text.encode(), b'GG' + b'GTACGGTCA')
    assert (h.row0, h.col0, h.start, h.end) == (7, 2, 5, 16)
    assert (h.matches, h.span) == (9, 11)
    # nothing in common
    assert flank.find_sequence(b'AAAAAAAA', b'CCCC').status == 1
    # scores with gap_open != gap_extend are refused
    assert flank.find_sequence(b'ACGT', b'ACGT', gap_open=-5, gap_extend=-1).status == 2


@pytest.mark.parametrize('seed', range(6))
def test_oracle_equals_literal_python(seed):
    rng = np.random.default_rng(seed)
    for _ in range(25):
        n, p = int(rng.integers(30, 400)), int(rng.integers(5, 70))
        text, pat = random_case(rng, n, p, float(rng.choice([0.0, 0.05, 0.15, 0.3])), edge=rng.choice([None, 'head', 'tail']))
        if not pat:
            continue
        want = py_find_sequence(text, pat)
        got = flank.find_sequence(text.encode(), pat.encode())
        if want is None:
            assert got.status == 1
            continue
        assert got.status == 0
        for k, v in want.items():
            assert getattr(got, k) == v, (k, getattr(got, k), v, text, pat)


def test_low_complexity_ties_follow_the_documented_rule():
    """Repeats make many equally good placements: the end cell is the LAST best cell in row-major order."""
    h = flank.find_sequence(b'ACACACACACAC', b'ACAC')
    assert (h.raw_score, h.row1, h.col1, h.row0) == (8, 12, 4, 8)
    w = py_find_sequence('ACACACACACAC', 'ACAC')
    assert (w['row1'], w['col1'], w['row0']) == (12, 4, 8)


def test_extract_from_moves_matches_reference_statement():
    rng = np.random.default_rng(3)
    for _ in range(20):
        m = (rng.random(int(rng.integers(5, 400))) < 0.4).astype(np.uint8)
        m[0] = 1
        # transform_moves (tr_extractor.py:147-163), literally
        mr = np.zeros(len(m), dtype=np.int32)
        for idx in range(1, len(m)):
            mr[idx] = mr[idx - 1] + (1 if m[idx] else 0)
        ps, pe = int(rng.integers(0, mr[-1] + 2)), int(rng.integers(0, mr[-1] + 3))
        d1, d2 = np.where(mr == ps)[0], np.where(mr == pe)[0]
        want = (100 + d1[0] * 5 if len(d1) else -1, 100 + d2[-1] * 5 if len(d2) else -1)
        assert flank.extract_from_moves(m, ps, pe, 100, 5) == want


def test_unique_optimum_is_pinned_and_ties_are_counted():
    """Bounding what the absent Biopython leaves unpinned: an independent NumPy Smith-Waterman (tests/sw_enumerate.py, no tie
    rule) enumerates every best end cell and every co-optimal traceback.  (i) Where the optimal alignment is unique, the
    oracle's hit equals the arithmetic of find_sequence on that one alignment, field for field -- tie rules cannot matter
    there, so this IS what pairwise2 returns; (ii) the hit says how much rests on a tie rule: n_best_cells = the number of
    best end cells, tie_steps = 0 exactly when the traceback from the chosen end cell is the only one."""
    from tests.sw_enumerate import hit_from_alignment, optimal_alignments
    rng = np.random.default_rng(77)
    unique = tied = 0
    cases = []
    for _ in range(120):                                    # short flanks in short texts: ties are common
        n, p = int(rng.integers(30, 300)), int(rng.integers(5, 40))
        cases.append(random_case(rng, n, p, float(rng.choice([0.0, 0.1, 0.25])), edge=rng.choice([None, 'head', 'tail'])))
    for _ in range(10):                                     # the shape upstream runs: 110-base flanks, thousands of bases
        n = int(rng.integers(2500, 5000))
        cases.append(random_case(rng, n, 110, float(rng.choice([0.08, 0.12]))))
    cases += [('ACGT' * 30, 'ACGTACGT'), ('ACACACACACAC', 'ACACGG'), ('TTTTTTTTTT', 'TTT')]  # periodic: many best cells
    for text, pat in cases:
        if not pat:
            continue
        best, ends, als = optimal_alignments(text, pat)
        h = flank.find_sequence(text.encode(), pat.encode())
        if best == 0:
            assert h.status == 1
            continue
        assert h.status == 0 and h.raw_score == best and h.n_best_cells == len(ends)
        assert (h.row1, h.col1) == max(ends)                # the documented rule: largest text index, then pattern index
        from_chosen_end = [a for a in als if (a[2], a[3]) == (h.row1, h.col1)]
        assert (h.tie_steps == 0) == (len(from_chosen_end) == 1)
        if len(als) < 64:                                    # (the enumeration stops spelling alignments out at 64)
            assert any(a[4].encode() == h.ops for a in from_chosen_end)  # the oracle's alignment is one of the optimal ones
        if len(ends) == 1 and len(als) == 1:
            unique += 1
            want = hit_from_alignment(text, pat, *als[0][:4], als[0][4], best)
            assert {k: getattr(h, k) for k in want} == want
            assert (h.n_best_cells, h.tie_steps) == (1, 0)
        else:
            tied += 1
    assert unique >= 40 and tied >= 10
