"""ctypes front-end of the CPU parity oracle (oracle/libwarpstr_oracle.so).

TEST INFRASTRUCTURE ONLY: import this from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never from warpstr_amd/.
"""
import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libwarpstr_oracle.so')

STATUS = {0: 'ok', 1: 'shape', 2: 'backtrack', 3: 'fit_points', 4: 'fit_order', 5: 'fit_smooth', 6: 'no_repeat',
          7: 'segment_range'}


class _Automaton(C.Structure):
    _fields_ = [('n_states', C.c_int32), ('endstate', C.c_int32), ('flank_length', C.c_int32),
                ('value', C.c_void_p), ('seq_idx', C.c_void_p), ('pred_ptr', C.c_void_p), ('pred_idx', C.c_void_p),
                ('repeat_mask', C.c_void_p)]


class _Params(C.Structure):
    _fields_ = [('min_values_per_state', C.c_int32), ('states_in_segment', C.c_int32), ('threshold', C.c_double),
                ('max_std', C.c_double), ('method_median', C.c_int32), ('reps_as_one', C.c_int32)]


class _Result(C.Structure):
    _fields_ = [('status', C.c_int32), ('len1', C.c_int32), ('len2', C.c_int32), ('n_trans1', C.c_int32),
                ('n_trans2', C.c_int32), ('cost1', C.c_double), ('cost2', C.c_double),
                ('dtw_end_cost1', C.c_double), ('dtw_end_cost2', C.c_double)]


class _Debug(C.Structure):
    _fields_ = [('trace1', C.c_void_p), ('trace2', C.c_void_p), ('rescaled', C.c_void_p), ('rescaled2', C.c_void_p),
                ('badmask', C.c_void_p), ('dlast1', C.c_void_p), ('dlast2', C.c_void_p), ('idx', C.c_int64 * 4), ('fit_knots', C.c_int64)]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, 'warpstr_oracle.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s', '-B', 'libwarpstr_oracle.so'])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.wso_np_mean.restype = C.c_double
        _lib.wso_np_std.restype = C.c_double
        _lib.wso_np_median.restype = C.c_double
        _lib.wso_transitions.restype = C.c_long
        _lib.wso_create_alignment.restype = C.c_long
        _lib.wso_segment.restype = C.c_long
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


@dataclass
class Params:
    min_values_per_state: int = 4
    states_in_segment: int = 6
    threshold: float = 0.5
    max_std: float = 0.5
    method: str = 'mean'
    reps_as_one: bool = False

    def c(self):
        return _Params(self.min_values_per_state, self.states_in_segment, self.threshold, self.max_std,
                       1 if self.method == 'median' else 0, 1 if self.reps_as_one else 0)


class Automaton:
    """Keeps the numpy buffers alive next to the C struct.  `table` is any object with the
    AutomatonTable fields (warpstr_amd.automata) -- plain arrays, no product code is called."""

    def __init__(self, value, seq_idx, pred_ptr, pred_idx, repeat_mask, endstate, flank_length):
        self.value = np.ascontiguousarray(value, dtype=np.float64)
        self.seq_idx = np.ascontiguousarray(seq_idx, dtype=np.int32)
        self.pred_ptr = np.ascontiguousarray(pred_ptr, dtype=np.int32)
        self.pred_idx = np.ascontiguousarray(pred_idx, dtype=np.int32)
        self.repeat_mask = np.ascontiguousarray(repeat_mask, dtype=np.uint8)
        self.endstate, self.flank_length = int(endstate), int(flank_length)
        self.n_states = len(self.value)
        self.c = _Automaton(self.n_states, self.endstate, self.flank_length, _p(self.value), _p(self.seq_idx),
                            _p(self.pred_ptr), _p(self.pred_idx), _p(self.repeat_mask))

    @classmethod
    def from_table(cls, t, flank_length):
        return cls(t.value, t.seq_idx, t.pred_ptr, t.pred_idx, t.repeat_mask, t.endstate, flank_length)


def np_mean(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return lib().wso_np_mean(_p(a), C.c_long(len(a)))


def np_std(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    tmp = np.empty_like(a)
    return lib().wso_np_std(_p(a), C.c_long(len(a)), _p(tmp))


def np_median(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    tmp = np.empty_like(a)
    return lib().wso_np_median(_p(a), C.c_long(len(a)), _p(tmp))


def dtw_fill(aut: Automaton, sig, mask=None, m=4):
    sig = np.ascontiguousarray(sig, dtype=np.float64)
    T = len(sig)
    D = np.empty((T, aut.n_states), dtype=np.float64)
    mk = np.ascontiguousarray(mask, dtype=np.uint8) if mask is not None else None
    rc = lib().wso_dtw_fill(C.byref(aut.c), _p(sig), C.c_long(T), _p(mk), C.c_int(m), _p(D))
    if rc:
        raise RuntimeError(f'oracle dtw_fill: {STATUS[rc]}')
    return D


def backtrack(aut: Automaton, D, sig, mask=None, m=4):
    sig = np.ascontiguousarray(sig, dtype=np.float64)
    T = len(sig)
    tr = np.empty(T, dtype=np.int32)
    mk = np.ascontiguousarray(mask, dtype=np.uint8) if mask is not None else None
    rc = lib().wso_backtrack(C.byref(aut.c), _p(D), _p(sig), C.c_long(T), _p(mk), C.c_int(m), _p(tr))
    if rc:
        raise RuntimeError(f'oracle backtrack: {STATUS[rc]}')
    return tr


def warp(aut: Automaton, sig, mask=None, m=4):
    """WarpSTR.warp: trace and terminal DP cost."""
    D = dtw_fill(aut, sig, mask, m)
    return backtrack(aut, D, sig, mask, m), float(D[-1, aut.endstate])


def fit_cubic(x, y):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    t, c, fp = np.empty(8), np.empty(4), C.c_double()
    rc = lib().wso_fit_cubic(_p(x), _p(y), C.c_long(len(x)), _p(t), _p(c), C.byref(fp))
    if rc:
        raise RuntimeError(f'oracle fit_cubic: {STATUS[rc]}')
    return t, c, fp.value


def eval_cubic(t, c, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    lib().wso_eval_cubic(_p(np.ascontiguousarray(t)), _p(np.ascontiguousarray(c)), _p(x), C.c_long(len(x)), _p(out))
    return out


def curfit(x, y, s=None):
    """splrep(x, y, s=s) in full (k = 3, unit weights): knots, coefficients (n each, the last four zero), fp, ier."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    m = len(x)
    t, c = np.zeros(m + 16), np.zeros(m + 16)
    n, fp, ier = C.c_int(), C.c_double(), C.c_int()
    rc = lib().wso_curfit(_p(x), _p(y), C.c_long(m), C.c_double(float(m if s is None else s)), _p(t), _p(c), C.byref(n),
                          C.byref(fp), C.byref(ier))
    if rc:
        raise RuntimeError(f'oracle curfit: {STATUS[rc]}')
    return t[:n.value].copy(), c[:n.value].copy(), fp.value, ier.value


def splev(t, c, x):
    """splev(x, (t, c, 3)), ext = 0."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    t = np.ascontiguousarray(t, dtype=np.float64)
    c = np.ascontiguousarray(c, dtype=np.float64)
    out = np.empty_like(x)
    lib().wso_splev(_p(t), C.c_int(len(t)), _p(c), _p(x), C.c_long(len(x)), _p(out))
    return out


def rescale_signal(sig, value, expected, good):
    """rescale_signal over alignment records (value, expected, good_enough) -- caller.py:304-318."""
    sig = np.ascontiguousarray(sig, dtype=np.float64)
    value = np.ascontiguousarray(value, dtype=np.float64)
    expected = np.ascontiguousarray(expected, dtype=np.float64)
    good = np.ascontiguousarray(good, dtype=np.uint8)
    out = np.empty_like(sig)
    rc = lib().wso_rescale_signal(_p(sig), C.c_long(len(sig)), _p(value), _p(expected), _p(good), C.c_long(len(value)), _p(out),
                                  None, None)
    if rc:
        raise RuntimeError(f'oracle rescale_signal: {STATUS[rc]}')
    return out


def segment(data, win=3):
    data = np.ascontiguousarray(data, dtype=np.float64)
    return lib().wso_segment(_p(data), C.c_long(len(data)), C.c_int(win))


@dataclass
class ReadCall:
    status: int
    len1: int
    len2: int
    n_trans1: int
    n_trans2: int
    cost1: float
    cost2: float
    dtw_end_cost1: float
    dtw_end_cost2: float
    trace1: Optional[np.ndarray] = None
    trace2: Optional[np.ndarray] = None
    rescaled: Optional[np.ndarray] = None
    rescaled2: Optional[np.ndarray] = None
    badmask: Optional[np.ndarray] = None
    dlast1: Optional[np.ndarray] = None
    dlast2: Optional[np.ndarray] = None
    idx: Optional[tuple] = None
    fit_knots: Optional[int] = None  # knots of the first pass's spline (8: the cubic; more: FITPACK's smoothing branch)


def call_read(aut: Automaton, sig, params: Params = Params(), debug: bool = True) -> ReadCall:
    """WarpSTR.run for one read (src/caller/caller.py:117-149)."""
    sig = np.ascontiguousarray(sig, dtype=np.float64)
    T = len(sig)
    res = _Result()
    pc = params.c()
    if debug:
        b = dict(trace1=np.zeros(T, np.int32), trace2=np.zeros(T, np.int32), rescaled=np.zeros(T), rescaled2=np.zeros(T),
                 badmask=np.zeros(T, np.uint8), dlast1=np.zeros(aut.n_states), dlast2=np.zeros(aut.n_states))
        dbg = _Debug(_p(b['trace1']), _p(b['trace2']), _p(b['rescaled']), _p(b['rescaled2']), _p(b['badmask']),
                     _p(b['dlast1']), _p(b['dlast2']))
        lib().wso_call_read(C.byref(aut.c), C.byref(pc), _p(sig), C.c_long(T), C.byref(res), C.byref(dbg))
        extra = dict(b, idx=tuple(dbg.idx), fit_knots=int(dbg.fit_knots))
    else:
        lib().wso_call_read(C.byref(aut.c), C.byref(pc), _p(sig), C.c_long(T), C.byref(res), None)
        extra = {}
    return ReadCall(res.status, res.len1, res.len2, res.n_trans1, res.n_trans2, res.cost1, res.cost2,
                    res.dtw_end_cost1, res.dtw_end_cost2, **extra)
