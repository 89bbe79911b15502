"""Flank localisation on the GPU: the reference's step-1 interface over the C ABI.

Mirrors (same names, argument meaning and result fields):
  Position, Mapping, Alignment, Flank       src/extractor/tr_extractor.py:25-73
  find_sequence(seq1, seq2, origin_offset)  tr_extractor.py:196-250   (here batched: find_sequences)
  align_seq(read, flank)                    tr_extractor.py:253-274   (here batched: align_seqs)
  transform_moves / extract_from_moves      tr_extractor.py:147-193   (extract_from_moves_batch)
All alignment arithmetic runs in HIP kernels (csrc/flank_kernels.hip); this module packs buffers and rebuilds the
`Mapping(ref, mapping, query)` strings from the returned operations.  Parity with Biopython's pairwise2 is UNPINNED
(see oracle/flank_oracle.c): where several optimal alignments exist the choice follows this library's documented rule.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib

ALIGNMENT_MATCH_CHAR, ALIGNMENT_MISMATCH_CHAR = '|', ' '   # src/templates.py


@dataclass
class AlignmentConfig:                                       # src/config.py:135-141
    accuracy_factor: float = 1.15
    identity_factor: float = 0.85
    match_score: int = 2
    mismatch_score: int = -3
    gap_open_score: int = -3
    gap_extend_score: int = -3


@dataclass
class Position:
    start: int
    end: int

    @property
    def valid(self) -> bool:
        return not (self.start > self.end or self.start == -1 or self.end == -1)


@dataclass
class Mapping:
    ref: str = '-'
    mapping: str = '-'
    query: str = '-'


@dataclass
class Alignment:
    score: int = -1
    identity: float = -1.0
    position: Position = field(default_factory=lambda: Position(-1, -1))
    mapping: Mapping = field(default_factory=Mapping)
    config: AlignmentConfig = field(default_factory=AlignmentConfig, repr=False)

    def __post_init__(self):
        if self.score <= self.config.accuracy_factor and self.identity <= self.config.identity_factor:
            self.position = Position(-1, -1)

    @property
    def found(self) -> bool:
        return self.position.valid


@dataclass
class Flank:
    left: str
    right: str


def _pack(seqs: Sequence[str]):
    lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=len(seqs))
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    buf = np.frombuffer(''.join(seqs).encode('ascii'), dtype=np.uint8) if off[-1] else np.zeros(0, np.uint8)
    return np.ascontiguousarray(buf), off


def locate(texts: Sequence[str], patterns: Sequence[str], config: Optional[AlignmentConfig] = None, device: int = 0):
    """Raw batch call: (hits as a structured array of _lib.FLANK_HIT_DTYPE, ops uint8[n, stride])."""
    cfg = config or AlignmentConfig()
    n = len(texts)
    assert len(patterns) == n
    text, toff = _pack(texts)
    pat, poff = _pack(patterns)
    hits = np.zeros(n, dtype=_lib.FLANK_HIT_DTYPE)
    stride = 2 * max((len(p) for p in patterns), default=0) + 8
    ops = np.zeros((n, stride), dtype=np.uint8)
    sc = _lib.WsxAlignScores(cfg.match_score, cfg.mismatch_score, cfg.gap_open_score, cfg.gap_extend_score)
    lib = _lib.load()
    _lib.check(lib.wsx_locate_flanks(device, None, _lib.WSX_MEM_HOST, _lib.ptr(text), _lib.ptr(toff), _lib.ptr(pat), _lib.ptr(poff),
                                     n, C.byref(sc), _lib.ptr(hits), _lib.ptr(ops), stride), 'wsx_locate_flanks')
    return hits, ops


def _mapping(text: str, pattern: str, h, ops: bytes) -> Mapping:
    """ref / mapping / query exactly as find_sequence slices them out of pairwise2's aligned strings (226-241)."""
    row0, col0, row1, col1 = int(h['row0']), int(h['col0']), int(h['row1']), int(h['col1'])
    al1 = ['-'] * max(col0 - row0, 0) + list(text[:row0])
    al2 = ['-'] * max(row0 - col0, 0) + list(pattern[:col0])
    ti, pj = row0, col0
    for op in ops[:int(h['n_ops'])]:
        if op != ord('L'):
            al1.append(text[ti]); ti += 1
        else:
            al1.append('-')
        if op != ord('U'):
            al2.append(pattern[pj]); pj += 1
        else:
            al2.append('-')
    st, sp = text[row1:], pattern[col1:]
    al1 += list(st) + ['-'] * max(len(sp) - len(st), 0)
    al2 += list(sp) + ['-'] * max(len(st) - len(sp), 0)
    lo, hi = int(h['start']), int(h['end'])
    ref, query = ''.join(al1[lo:hi]), ''.join(al2[lo:hi])
    mapping = ''.join(ALIGNMENT_MATCH_CHAR if a == b else ALIGNMENT_MISMATCH_CHAR for a, b in zip(ref, query))
    return Mapping(ref=ref, mapping=mapping, query=query)


def find_sequences(texts: Sequence[str], patterns: Sequence[str], origin_offsets: Optional[Sequence[int]] = None,
                   config: Optional[AlignmentConfig] = None, with_mapping: bool = True, device: int = 0) -> List[Alignment]:
    """find_sequence for a batch.  A pair without any positive-scoring alignment makes upstream raise (IndexError on
    pairwise2's empty list); here it yields the default Alignment() (not found)."""
    cfg = config or AlignmentConfig()
    hits, ops = locate(texts, patterns, cfg, device)
    out = []
    for r in range(len(texts)):
        h = hits[r]
        if h['status'] != 0:
            out.append(Alignment(config=cfg))
            continue
        off = int(origin_offsets[r]) if origin_offsets is not None else 0
        span = int(h['span'])
        ident = int(h['matches']) / span if span else 0.0
        mp = _mapping(texts[r], patterns[r], h, ops[r].tobytes()) if with_mapping else Mapping()
        out.append(Alignment(score=int(h['score']), identity=ident, position=Position(int(h['start']) + off, int(h['end']) + off),
                             mapping=mp, config=cfg))
    return out


def align_seqs(reads: Sequence[str], flanks: Sequence[Flank], config: Optional[AlignmentConfig] = None, device: int = 0
               ) -> List[Tuple[Alignment, Alignment]]:
    """align_seq for a batch: the left flank in the whole read, then the right flank in the part after the left hit."""
    cfg = config or AlignmentConfig()
    left = find_sequences(reads, [f.left for f in flanks], None, cfg, device=device)
    idx, texts, pats, offs = [], [], [], []
    for r, (read, fl, la) in enumerate(zip(reads, flanks, left)):
        if la.position.end + len(fl.right) > len(read):     # tr_extractor.py:268-269 (also for a left flank not found: -1)
            continue
        idx.append(r)
        texts.append(read[la.position.end:])
        pats.append(fl.right)
        offs.append(la.position.end)
    right_found = find_sequences(texts, pats, offs, cfg, device=device) if idx else []
    right = [Alignment(config=cfg) for _ in reads]
    for r, al in zip(idx, right_found):
        right[r] = al
    return list(zip(left, right))


def extract_from_moves_batch(moves: Sequence[np.ndarray], positions: Sequence[Position], strand_starts: Sequence[int],
                             block_strides: Sequence[int], device: int = 0) -> List[Position]:
    """transform_moves + extract_from_moves for a batch of reads (one Position per read)."""
    n = len(moves)
    lens = np.fromiter((len(m) for m in moves), dtype=np.int64, count=n)
    off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    buf = np.ascontiguousarray(np.concatenate([np.asarray(m, dtype=np.uint8) for m in moves])) if n else np.zeros(0, np.uint8)
    ps = np.array([p.start for p in positions], dtype=np.int32)
    pe = np.array([p.end for p in positions], dtype=np.int32)
    ss = np.array(strand_starts, dtype=np.int64)
    bs = np.array(block_strides, dtype=np.int32)
    rs, re = np.zeros(n, np.int64), np.zeros(n, np.int64)
    lib = _lib.load()
    _lib.check(lib.wsx_moves_to_raw(device, None, _lib.WSX_MEM_HOST, _lib.ptr(buf), _lib.ptr(off), _lib.ptr(ps), _lib.ptr(pe),
                                    _lib.ptr(ss), _lib.ptr(bs), n, _lib.ptr(rs), _lib.ptr(re)), 'wsx_moves_to_raw')
    return [Position(int(a), int(b)) for a, b in zip(rs, re)]


@dataclass
class BasecalledRead:
    """What extract_tr reads from an annotated fast5 (Fast5.get_tr_extract_reqs, src/schemas/fast5.py:35-45)."""
    name: str
    reverse: bool
    fasta: str
    moves: np.ndarray
    strand_start: int
    block_stride: int
    approx_location: Optional[int] = None


@dataclass
class FlankInRead:                                           # tr_extractor.py:76-100
    read_id: str
    lflank_raw: Position
    rflank_raw: Position
    l_alignment: Alignment
    r_alignment: Alignment
    sequence: Optional[str]

    @property
    def valid(self) -> int:
        return int(self.l_alignment.found and self.r_alignment.found and self.lflank_raw.valid and self.rflank_raw.valid)


_COMPLEMENT = str.maketrans('ACGTacgt', 'TGCAtgca')


def extract_tr_batch(reads: Sequence[BasecalledRead], template: Flank, reverse: Flank,
                     config: Optional[AlignmentConfig] = None, device: int = 0) -> List[FlankInRead]:
    """extract_tr (tr_extractor.py:277-340) for a batch of basecalled reads, without the file handling: the search
    window around the mapped location, both flank alignments, the move-table mapping to raw positions and the
    basecalled repeat sequence."""
    cfg = config or AlignmentConfig()
    windows, starts, flanks = [], [], []
    for rd in reads:
        n = len(rd.fasta)
        if not rd.approx_location:
            start, end = 0, n - 1
        else:
            loc = min(max(rd.approx_location, 0), n)
            start = max(int(loc - 0.05 * n - 5000), 0)
            end = min(int(loc + 0.05 * n + 5000), n - 1)
        windows.append(rd.fasta[start:end])
        starts.append(start)
        flanks.append(reverse if rd.reverse else template)
    pairs = align_seqs(windows, flanks, cfg, device=device)
    out: List[Optional[FlankInRead]] = [None] * len(reads)
    idx, mv, pos_l, pos_r, ss, bs = [], [], [], [], [], []
    for r, (rd, (la, ra)) in enumerate(zip(reads, pairs)):
        if not la.found or not ra.found:
            out[r] = FlankInRead(rd.name, Position(-1, -1), Position(-1, -1), la, ra, None)
            continue
        for al in (la, ra):
            al.position.start += starts[r]
            al.position.end += starts[r]
        idx.append(r)
        mv += [rd.moves, rd.moves]
        pos_l.append(la.position)
        pos_r.append(ra.position)
        ss += [rd.strand_start, rd.strand_start]
        bs += [rd.block_stride, rd.block_stride]
    if idx:
        raw = extract_from_moves_batch(mv, [p for pair in zip(pos_l, pos_r) for p in pair], ss, bs, device=device)
        for k, r in enumerate(idx):
            rd, (la, ra) = reads[r], pairs[r]
            seq = rd.fasta[la.position.end:ra.position.start]
            if seq and rd.reverse:
                seq = seq.translate(_COMPLEMENT)[::-1]       # get_reverse_strand
            out[r] = FlankInRead(rd.name, raw[2 * k], raw[2 * k + 1], la, ra, seq)
    return out  # type: ignore[return-value]
