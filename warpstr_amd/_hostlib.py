"""ctypes binding of warpstr_amd/_host_loci.so (csrc/host_loci.cpp): the per-locus host work of step 3 -- overview.csv in and
out, the two automata, FASTA files, the complex-unit table -- as native code that runs without the GIL.  The Python forms
(automata.py, overview.py, units.py) are the definition; every entry here either reproduces them byte for byte or declines
(returns None), and the caller then runs the Python form.  WARPSTR_NO_HOST_NATIVE=1 turns the library off."""
import ctypes as C
import os
from typing import Optional

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_host_loci.so')
_LIB = False


class WshAutomaton(C.Structure):
    _fields_ = [('n_states', C.c_int32), ('endstate', C.c_int32), ('repstart', C.c_int32), ('repend', C.c_int32),
                ('n_edges', C.c_int32), ('reserved', C.c_int32), ('value', C.c_void_p), ('seq_idx', C.c_void_p),
                ('pred_ptr', C.c_void_p), ('pred_idx', C.c_void_p), ('repeat_mask', C.c_void_p), ('last_base', C.c_void_p),
                ('kmer', C.c_void_p), ('owner', C.c_void_p)]


class WshOverviewInfo(C.Structure):
    _fields_ = [('n_rows', C.c_int32), ('n_cols', C.c_int32), ('n_saved', C.c_int32), ('has_run_id', C.c_int32),
                ('has_fast5_path', C.c_int32), ('reserved', C.c_int32), ('saved_rows', C.c_void_p), ('reverse', C.c_void_p),
                ('lo', C.c_void_p), ('hi', C.c_void_p), ('names', C.c_void_p), ('names_off', C.c_void_p),
                ('runs', C.c_void_p), ('runs_off', C.c_void_p), ('f5s', C.c_void_p), ('f5s_off', C.c_void_p)]


EXPORTS = ['wsh_abi_version', 'wsh_format_float', 'wsh_automaton_compile', 'wsh_automaton_free', 'wsh_locus_open', 'wsh_locus_error',
           'wsh_locus_free', 'wsh_locus_info', 'wsh_locus_text', 'wsh_locus_store', 'wsh_free', 'wsh_collapse_store']


def lib():
    """The library, or None (not built, switched off)."""
    global _LIB
    if _LIB is False:
        _LIB = None
        if os.path.exists(_PATH) and not os.environ.get('WARPSTR_NO_HOST_NATIVE'):
            try:
                h = C.CDLL(_PATH)
                if h.wsh_abi_version() == 1:
                    h.wsh_format_float.argtypes = [C.c_double, C.c_char_p]
                    h.wsh_automaton_compile.argtypes = [C.c_char_p, C.c_int64, C.c_void_p, C.c_int32, C.POINTER(WshAutomaton)]
                    h.wsh_automaton_free.argtypes = [C.POINTER(WshAutomaton)]
                    h.wsh_automaton_free.restype = None
                    h.wsh_locus_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
                    h.wsh_locus_error.argtypes = [C.c_void_p]
                    h.wsh_locus_error.restype = C.c_char_p
                    h.wsh_locus_free.argtypes = [C.c_void_p]
                    h.wsh_locus_free.restype = None
                    h.wsh_locus_info.argtypes = [C.c_void_p, C.POINTER(WshOverviewInfo)]
                    h.wsh_locus_info.restype = None
                    h.wsh_locus_text.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
                    h.wsh_locus_text.restype = C.c_void_p
                    h.wsh_locus_store.argtypes = [C.c_void_p, C.c_char_p] + [C.c_void_p] * 6 + [C.c_int32, C.POINTER(C.c_void_p),
                                                                                                C.POINTER(C.c_int64)]
                    h.wsh_free.argtypes = [C.c_void_p]
                    h.wsh_free.restype = None
                    h.wsh_collapse_store.argtypes = [C.c_char_p, C.c_int32] + [C.c_void_p] * 4 + [C.c_int32, C.c_void_p, C.c_char_p,
                                                                                                   C.c_void_p, C.c_void_p, C.c_char_p, C.c_int32,
                                                                                                   C.c_void_p, C.POINTER(C.c_void_p),
                                                                                                   C.POINTER(C.c_int64)]
                    _LIB = h
            except OSError:
                pass
    return _LIB


def _copy(ptr, dtype, count):
    if count == 0 or not ptr:
        return np.zeros(0, dtype)
    n = count * np.dtype(dtype).itemsize
    return np.frombuffer(bytearray(C.string_at(ptr, n)), dtype=dtype)  # (a writable copy)


def format_float(x: float) -> str:
    buf = C.create_string_buffer(48)
    n = lib().wsh_format_float(float(x), buf)
    return buf.raw[:n].decode('ascii')


def compile_automaton(pattern: str, pore_model):
    """AutomatonTable of `pattern` (automata.compile_automaton's result, field for field), or None when the library is absent
    or leaves the pattern to the Python compiler (which raises on it)."""
    h = lib()
    if h is None:
        return None
    try:
        pat = pattern.encode('ascii')
    except UnicodeEncodeError:
        return None
    levels = pore_model.level_norm
    if levels.dtype != np.float64 or not levels.flags.c_contiguous:
        levels = np.ascontiguousarray(levels, np.float64)
    a = WshAutomaton()
    if h.wsh_automaton_compile(pat, len(pat), levels.ctypes.data, int(pore_model.kmersize), C.byref(a)) != 0:
        return None
    try:
        from .automata import AutomatonTable
        S, E = a.n_states, a.n_edges
        return AutomatonTable(n_states=S, endstate=a.endstate, value=_copy(a.value, np.float64, S), seq_idx=_copy(a.seq_idx, np.int32, S),
                              pred_ptr=_copy(a.pred_ptr, np.int32, S + 1), pred_idx=_copy(a.pred_idx, np.int32, E),
                              repeat_mask=_copy(a.repeat_mask, np.uint8, S), last_base=_copy(a.last_base, np.uint8, S),
                              kmer_codes=_copy(a.kmer, np.uint32, S), kmersize=int(pore_model.kmersize), repstart=a.repstart,
                              repend=a.repend)
    finally:
        h.wsh_automaton_free(C.byref(a))


class NativeOverview:
    """overview.csv of one locus, parsed by the library: the `saved` rows' columns as arrays, and later the writer of the locus's
    step-3 files.  open() returns None when the table has to go through pandas (reason in `NativeOverview.last_refusal`)."""
    last_refusal = ''

    def __init__(self, handle, path):
        self._h, self.path = handle, path
        info = WshOverviewInfo()
        lib().wsh_locus_info(handle, C.byref(info))
        n = self.n_saved = info.n_saved
        self.n_rows = info.n_rows
        self.saved = _copy(info.saved_rows, np.int32, n).astype(np.int64)
        self.reverse = _copy(info.reverse, np.uint8, n).astype(bool)
        self.lo, self.hi = _copy(info.lo, np.int64, n), _copy(info.hi, np.int64, n)
        self._names = self._split(info.names, info.names_off, n)
        self.run_id = self._split(info.runs, info.runs_off, n) if info.has_run_id else None
        self.fast5_path = self._split(info.f5s, info.f5s_off, n) if info.has_fast5_path else None

    @staticmethod
    def _split(blob, off, n):
        off = _copy(off, np.int64, n + 1)
        text = C.string_at(blob, int(off[-1])).decode('utf-8') if n else ''
        if len(text) != int(off[-1]):  # (non-ASCII names: offsets are bytes)
            raw = C.string_at(blob, int(off[-1]))
            return [raw[a:b].decode('utf-8') for a, b in zip(off[:-1].tolist(), off[1:].tolist())]
        o = off.tolist()
        return [text[o[i]:o[i + 1]] for i in range(n)]

    @property
    def names(self):
        return self._names

    @classmethod
    def open(cls, overview_path: str) -> Optional['NativeOverview']:
        h = lib()
        if h is None:
            return None
        handle = C.c_void_p()
        rc = h.wsh_locus_open(os.fsencode(overview_path), C.byref(handle))
        if rc != 0:
            cls.last_refusal = (h.wsh_locus_error(handle) or b'').decode('utf-8', 'replace')
            h.wsh_locus_free(handle)
            if rc < 0:
                raise FileNotFoundError(f'Not found the overview file {overview_path} - Please check the "output" in config')
            return None
        return cls(handle, overview_path)

    def text(self) -> str:
        n = C.c_int64()
        p = lib().wsh_locus_text(self._h, C.byref(n))
        return C.string_at(p, n.value).decode('utf-8')

    def store(self, locus_path: str, len1, len2, cost1, cost2, seq2, off2, write: bool) -> str:
        """Writes overview.csv and the FASTA files (write=True) and returns the new overview table as CSV text."""
        h = lib()
        arrs = [np.ascontiguousarray(len1, np.int32), np.ascontiguousarray(len2, np.int32), np.ascontiguousarray(cost1, np.float64),
                np.ascontiguousarray(cost2, np.float64), np.ascontiguousarray(seq2, np.uint8), np.ascontiguousarray(off2, np.int64)]
        if any(len(a) != self.n_saved for a in arrs[:4] + arrs[5:]):
            raise ValueError(f'{self.n_saved} saved reads in the overview but {len(arrs[0])} results')
        out, n = C.c_void_p(), C.c_int64()
        rc = h.wsh_locus_store(self._h, os.fsencode(locus_path), *[a.ctypes.data for a in arrs], 3 if write else 0, C.byref(out), C.byref(n))
        if rc != 0:
            raise OSError((h.wsh_locus_error(self._h) or b'').decode('utf-8', 'replace'))
        try:
            return C.string_at(out, n.value).decode('utf-8')
        finally:
            h.wsh_free(out)

    def close(self):
        if self._h is not None and lib() is not None:
            lib().wsh_locus_free(self._h)
        self._h = None

    __del__ = close


def collapse_store(locus_path: str, seq2, off2, len2, reverse, repeat_units, offsets, header: str, write: bool):
    """collapse_repeats for every called sequence + the complex-unit table: (counts [n x alternatives], CSV text), or None when
    the Python form has to run (library absent, an empty alternative)."""
    h = lib()
    if h is None:
        return None
    n = len(len2)
    n_alt = np.array([len(u) for u in repeat_units], np.int32)
    flat = [a.encode('ascii') for u in repeat_units for a in u]
    alt_off = np.zeros(len(flat) + 1, np.int32)
    np.cumsum([len(a) for a in flat], out=alt_off[1:])
    blob = b''.join(flat)
    counts = np.zeros((n, len(flat)), np.int64)
    arrs = [np.ascontiguousarray(seq2, np.uint8), np.ascontiguousarray(off2, np.int64), np.ascontiguousarray(len2, np.int32),
            np.ascontiguousarray(reverse, np.uint8)]
    offs = np.ascontiguousarray(offsets, np.int32)
    out, ln = C.c_void_p(), C.c_int64()
    rc = h.wsh_collapse_store(os.fsencode(locus_path), n, *[a.ctypes.data for a in arrs], len(repeat_units), n_alt.ctypes.data, blob,
                              alt_off.ctypes.data, offs.ctypes.data, header.encode('utf-8'), 1 if write else 0, counts.ctypes.data,
                              C.byref(out), C.byref(ln))
    if rc > 0:
        return None
    if rc < 0:
        raise OSError(f'cannot write the complex-unit table under {locus_path}')
    try:
        return counts, C.string_at(out, ln.value).decode('utf-8')
    finally:
        h.wsh_free(out)
