"""Flank localisation on the GPU through the C ABI (wsx_locate_flanks, wsx_moves_to_raw) against the CPU
oracle: every field of every hit and the operation strings are compared exactly."""
import numpy as np
import pytest

from oracle import flank
from tests.test_flank_oracle import mutate, py_find_sequence, random_case
from warpstr_amd import _lib, extractor

pytestmark = pytest.mark.gpu


def _check(texts, pats):
    hits, ops = extractor.locate(texts, pats)
    for r, (t, p) in enumerate(zip(texts, pats)):
        o = flank.find_sequence(t.encode(), p.encode())
        h = hits[r]
        assert int(h['status']) == o.status, (r, t[:40], p)
        if o.status:
            continue
        for k in _lib.FLANK_HIT_DTYPE.names:
            assert int(h[k]) == getattr(o, k), (r, k, int(h[k]), getattr(o, k), p)
        assert ops[r, :o.n_ops].tobytes() == o.ops and not ops[r, o.n_ops:].any()
    return hits


@pytest.mark.parametrize('plen,nmax', [((5, 64), 600), ((65, 128), 3000), ((129, 256), 2000)])
def test_locate_matches_oracle(plen, nmax):
    rng = np.random.default_rng(plen[0])
    texts, pats = [], []
    for _ in range(60):
        n, p = int(rng.integers(plen[1] + 10, nmax)), int(rng.integers(plen[0], plen[1] + 1))
        t, q = random_case(rng, n, p, float(rng.choice([0.0, 0.05, 0.12, 0.25])), edge=rng.choice([None, None, 'head', 'tail']))
        if len(q) < plen[0]:
            q = q + t[:plen[0] - len(q)]
        texts.append(t)
        pats.append(q[:256])
    texts += ['A' * 50, 'ACGT' * 30, '', 'ACACACACACAC']                  # no hit, periodic ties, empty text, ties
    pats += ['C' * max(plen[0], 8), 'ACGTACGT' + 'T' * max(plen[0] - 8, 0), 'ACGT' * max(plen[0] // 4, 2), 'ACAC' + 'G' * max(plen[0] - 4, 1)]
    hits = _check(texts, pats)
    assert (hits['status'] == 0).sum() >= 55


def test_upstream_shaped_batch():
    """The shape upstream runs: 110-base flanks (flank_length default) in windows of ~10-13 k basecalled bases
    (extract_tr: 5 % of the read +- 5000 around the mapped location), Guppy-like error rates, both flanks per read."""
    rng = np.random.default_rng(2024)
    reads, flanks = [], []
    for _ in range(12):
        n = int(rng.integers(9000, 14000))
        read = ''.join('ACGT'[k] for k in rng.integers(0, 4, size=n))
        a = int(rng.integers(200, n - 1500))
        rep = 'AGC' * int(rng.integers(5, 60))
        read = read[:a + 110] + rep + read[a + 110:]
        left, right = mutate(rng, read[a:a + 110], 0.08), mutate(rng, read[a + 110 + len(rep):a + 220 + len(rep)], 0.08)
        reads.append(read)
        flanks.append(extractor.Flank(left=left[:110], right=right[:110]))
    pairs = extractor.align_seqs(reads, flanks)
    for read, fl, (la, ra) in zip(reads, flanks, pairs):
        o = flank.find_sequence(read.encode(), fl.left.encode())
        assert la.found and (la.position.start, la.position.end, la.score) == (o.start, o.end, o.score)
        assert abs(la.identity - o.matches / o.span) < 1e-15
        w = py_find_sequence(read, fl.left)
        assert (w['start'], w['end'], w['score'], w['matches']) == (o.start, o.end, o.score, o.matches)
        assert len(la.mapping.ref) == len(la.mapping.query) == len(la.mapping.mapping) == o.span
        assert la.mapping.mapping.count('|') == o.matches
        assert la.mapping.query.replace('-', '') in fl.left and la.mapping.ref.replace('-', '') in read
        # the right flank is searched after the left one and reported in read coordinates
        o2 = flank.find_sequence(read[la.position.end:].encode(), fl.right.encode())
        assert ra.found and (ra.position.start, ra.position.end) == (o2.start + la.position.end, o2.end + la.position.end)
        assert ra.position.start >= la.position.end
    # a flank that is not in the read at all: low score AND low identity -> position (-1, -1), as Alignment.__post_init__
    miss = extractor.find_sequences(['ACGT' * 100], ['TTTTTTTTTTGGGGGGGGGGCCCCCCCCCCAAAAAAAAAATTTTTTTTTTGGGGGGGGGGCCCCCCCCCC'])[0]
    assert miss.score > 1.15 or not miss.found


def test_moves_to_raw_matches_oracle():
    rng = np.random.default_rng(9)
    moves, pos, ss, bs = [], [], [], []
    for _ in range(40):
        m = (rng.random(int(rng.integers(1, 60000))) < 0.45).astype(np.uint8)
        m[0] = 1
        total = int(m[1:].sum())
        moves.append(m)
        pos.append(extractor.Position(int(rng.integers(0, total + 2)), int(rng.integers(0, total + 3))))
        ss.append(int(rng.integers(0, 5000)))
        bs.append(int(rng.choice([5, 10])))
    got = extractor.extract_from_moves_batch(moves, pos, ss, bs)
    for m, p, s, b, g in zip(moves, pos, ss, bs, got):
        assert (g.start, g.end) == flank.extract_from_moves(m, p.start, p.end, s, b)


def test_extract_tr_batch_end_to_end():
    """extract_tr's flow on synthetic basecalled reads: window around the mapped location, both flanks, raw positions
    through the move table, repeat sequence (reverse-complemented for reverse-strand reads)."""
    rng = np.random.default_rng(31)
    tpl = extractor.Flank(left=''.join('ACGT'[k] for k in rng.integers(0, 4, 110)), right=''.join('ACGT'[k] for k in rng.integers(0, 4, 110)))
    comp = str.maketrans('ACGT', 'TGCA')
    rev = extractor.Flank(left=tpl.right.translate(comp)[::-1], right=tpl.left.translate(comp)[::-1])
    reads, truth = [], []
    for k in range(10):
        is_rev = bool(k % 2)
        fl = rev if is_rev else tpl
        n0 = int(rng.integers(2000, 60000))
        head = ''.join('ACGT'[q] for q in rng.integers(0, 4, n0))
        rep = 'AGC' * int(rng.integers(8, 40))
        tail = ''.join('ACGT'[q] for q in rng.integers(0, 4, int(rng.integers(500, 8000))))
        fasta = head + fl.left + rep + fl.right + tail
        stride = 5
        moves = np.zeros(len(fasta) * 2 + 3, dtype=np.uint8)   # every base lasts two blocks
        moves[::2][:len(fasta)] = 1
        reads.append(extractor.BasecalledRead(f'r{k}', is_rev, fasta, moves, strand_start=100 + k, block_stride=stride,
                                              approx_location=n0 + 50 if k % 3 else None))
        truth.append((n0, n0 + 110, n0 + 110 + len(rep), n0 + 220 + len(rep), rep))
    res = extractor.extract_tr_batch(reads, tpl, rev)
    for rd, r, (ls, le, rs_, re_, rep) in zip(reads, res, truth):
        assert r.valid == 1 and r.read_id == rd.name
        assert (r.l_alignment.position.start, r.l_alignment.position.end) == (ls, le)
        assert (r.r_alignment.position.start, r.r_alignment.position.end) == (rs_, re_)
        assert r.l_alignment.identity == 1.0 and r.l_alignment.score == 220
        want_seq = rep.translate(comp)[::-1] if rd.reverse else rep
        assert r.sequence == want_seq
        # base b occupies blocks 2b and 2b+1: context index b first appears at block 2b, last at 2b+1
        assert r.lflank_raw.start == rd.strand_start + 2 * ls * rd.block_stride
        assert r.lflank_raw.end == rd.strand_start + (2 * le + 1) * rd.block_stride
        assert r.rflank_raw.start == rd.strand_start + 2 * rs_ * rd.block_stride
    # a read that does not contain the flanks
    junk = extractor.BasecalledRead('junk', False, 'ACGT' * 500, np.ones(100, np.uint8), 0, 5)
    bad = extractor.extract_tr_batch([junk], tpl, rev)[0]
    assert bad.valid == 0 and bad.sequence is None and bad.lflank_raw.start == -1
