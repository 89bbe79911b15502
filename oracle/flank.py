"""ctypes front-end of the flank-localisation oracle (oracle/libflank_oracle.so, flank_oracle.c).

TEST INFRASTRUCTURE ONLY: import this from tests/ (and scripts that check the HIP path), never from warpstr_amd/.
"""
import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libflank_oracle.so')


class _Scores(C.Structure):
    _fields_ = [('match', C.c_int32), ('mismatch', C.c_int32), ('gap_open', C.c_int32), ('gap_extend', C.c_int32)]


class _Hit(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('status', 'score', 'start', 'end', 'matches', 'span', 'row0', 'col0', 'row1', 'col1',
                                         'gaps_text', 'gaps_pattern', 'raw_score', 'n_ops', 'n_best_cells', 'tie_steps')]


HIT_FIELDS = [n for n, _ in _Hit._fields_]
_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, 'flank_oracle.c')
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.check_call(['make', '-C', _HERE, '-s', '-B', 'libflank_oracle.so'])
        _lib = C.CDLL(_SO)
        _lib.flo_find_sequence.restype = C.c_int
        _lib.flo_window_rows.restype = C.c_int32
    return _lib


@dataclass
class Hit:
    status: int
    score: int
    start: int
    end: int
    matches: int
    span: int
    row0: int
    col0: int
    row1: int
    col1: int
    gaps_text: int
    gaps_pattern: int
    raw_score: int
    n_ops: int
    n_best_cells: int
    tie_steps: int
    ops: bytes


def find_sequence(text: bytes, pattern: bytes, match=2, mismatch=-3, gap_open=-3, gap_extend=-3) -> Hit:
    t = np.frombuffer(text, dtype=np.uint8)
    p = np.frombuffer(pattern, dtype=np.uint8)
    sc = _Scores(match, mismatch, gap_open, gap_extend)
    hit = _Hit()
    cap = 2 * len(p) + 8
    ops = np.zeros(cap, dtype=np.uint8)
    lib().flo_find_sequence(t.ctypes.data_as(C.c_void_p), C.c_int32(len(t)), p.ctypes.data_as(C.c_void_p), C.c_int32(len(p)),
                            C.byref(sc), C.byref(hit), ops.ctypes.data_as(C.c_void_p), C.c_int32(cap))
    vals = {n: int(getattr(hit, n)) for n in HIT_FIELDS}
    return Hit(ops=ops[:min(vals['n_ops'], cap)].tobytes(), **vals)


def extract_from_moves(moves: np.ndarray, pos_start: int, pos_end: int, strand_start: int, block_stride: int):
    m = np.ascontiguousarray(moves, dtype=np.uint8)
    a, b = C.c_int64(), C.c_int64()
    lib().flo_extract_from_moves(m.ctypes.data_as(C.c_void_p), C.c_int64(len(m)), C.c_int32(pos_start), C.c_int32(pos_end),
                                 C.c_int64(strand_start), C.c_int32(block_stride), C.byref(a), C.byref(b))
    return int(a.value), int(b.value)
