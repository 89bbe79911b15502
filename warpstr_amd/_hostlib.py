"""ctypes binding of warpstr_amd/_host_loci.so (csrc/host_loci.cpp): the per-locus host work of step 3 -- overview.csv in and
out, the two automata, FASTA files, the complex-unit table -- as native code that runs without the GIL.  The Python forms
(automata.py, overview.py, units.py) are the definition; every entry here either reproduces them byte for byte or declines
(returns None), and the caller then runs the Python form.  WARPSTR_NO_HOST_NATIVE=1 turns the library off."""
import collections.abc
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


_AUT_DTYPE = np.dtype([(k, np.int32) for k in ('n_states', 'endstate', 'repstart', 'repend', 'n_edges', 'reserved')]
                      + [(k, np.uint64) for k in ('value', 'seq_idx', 'pred_ptr', 'pred_idx', 'repeat_mask', 'last_base', 'kmer', 'owner')])
_SETUP_DTYPE = np.dtype([('owner', np.uint64), ('locus', np.uint64), ('overview_status', np.int32), ('automata_status', np.int32),
                         ('similarity_status', np.int32), ('reserved', np.int32), ('aut', _AUT_DTYPE, (2,)), ('similarity_csv', np.uint64),
                         ('warnings', np.uint64)])   # WshSetup, field for field (checked when the library is loaded)


class WshSetup(C.Structure):
    _fields_ = [('owner', C.c_void_p), ('locus', C.c_void_p), ('overview_status', C.c_int32), ('automata_status', C.c_int32),
                ('similarity_status', C.c_int32), ('reserved', C.c_int32), ('aut', WshAutomaton * 2), ('similarity_csv', C.c_char_p),
                ('warnings', C.c_char_p)]


WSH_ABI = 2   # bumped whenever the exports or their meaning change (csrc/host_loci.cpp: wsh_abi_version)
assert _SETUP_DTYPE.itemsize == C.sizeof(WshSetup) and _AUT_DTYPE.itemsize == C.sizeof(WshAutomaton)
assert all(_SETUP_DTYPE.fields[k][1] == getattr(WshSetup, k).offset for k in _SETUP_DTYPE.names)
assert all(_AUT_DTYPE.fields[k][1] == getattr(WshAutomaton, k).offset for k in _AUT_DTYPE.names)

EXPORTS = ['wsh_abi_version', 'wsh_format_float', 'wsh_automaton_compile', 'wsh_automaton_free', 'wsh_locus_open', 'wsh_locus_error',
           'wsh_locus_free', 'wsh_locus_info', 'wsh_locus_text', 'wsh_locus_store', 'wsh_free', 'wsh_collapse_store', 'wsh_locus_setup',
           'wsh_setup_free', 'wsh_locus_table', 'wsh_loci_store', 'wsh_loci_setup', 'wsh_vbz_decode_i16', 'wsh_vbz_unpack', 'wsh_vbz_context', 'wsh_gather', 'wsh_loci_counts', 'wsh_loci_columns']


def lib():
    """The library, or None (not built, switched off)."""
    global _LIB
    if _LIB is False:
        _LIB = None
        if os.path.exists(_PATH) and not os.environ.get('WARPSTR_NO_HOST_NATIVE'):
            try:
                h = C.CDLL(_PATH)
                if h.wsh_abi_version() == WSH_ABI:
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
                    h.wsh_locus_store.argtypes = [C.c_void_p, C.c_char_p] + [C.c_void_p] * 6 + [C.c_int32]
                    h.wsh_locus_table.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
                    h.wsh_locus_table.restype = C.c_void_p
                    h.wsh_loci_store.argtypes = [C.c_int32] + [C.c_void_p] * 9 + [C.c_int32, C.c_void_p]
                    h.wsh_loci_setup.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_int32, C.c_void_p]
                    h.wsh_loci_setup.restype = None
                    h.wsh_loci_counts.argtypes = [C.c_int32] + [C.c_void_p] * 5
                    h.wsh_loci_counts.restype = None
                    h.wsh_loci_columns.argtypes = [C.c_int32] + [C.c_void_p] * 11
                    h.wsh_loci_columns.restype = None
                    h.wsh_free.argtypes = [C.c_void_p]
                    h.wsh_free.restype = None
                    h.wsh_collapse_store.argtypes = [C.c_char_p, C.c_int32] + [C.c_void_p] * 4 + [C.c_int32, C.c_void_p, C.c_char_p,
                                                                                                   C.c_void_p, C.c_void_p, C.c_char_p, C.c_int32, C.c_void_p,
                                                                                                   C.c_int32, C.c_void_p, C.POINTER(C.c_void_p),
                                                                                                   C.POINTER(C.c_int64)]
                    h.wsh_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32]
                    h.wsh_gather.restype = None
                    h.wsh_locus_setup.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_int32, C.c_double, C.c_int32, C.POINTER(WshSetup)]
                    h.wsh_locus_setup.restype = None
                    h.wsh_setup_free.argtypes = [C.POINTER(WshSetup)]
                    h.wsh_setup_free.restype = None
                    _LIB = h
            except (OSError, AttributeError):   # (a stale build that lacks an export is no library at all)
                _LIB = None
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


def _under(directory: str, name: str) -> str:
    """os.path.join(directory, name) for a plain file name (a tenth of its time: this runs per locus under the interpreter's lock)."""
    return directory + name if directory.endswith(os.sep) else directory + os.sep + name if directory else name


class _Strings(collections.abc.Sequence):
    """Rows [a, b) of a string column that a chunk of loci handed over in one piece (text + the rows' offsets): the strings are made
    when somebody looks at them -- a run of thousands of loci looks at a read's name when it hands the read to a reader, thirty
    strings per locus made under the interpreter's lock were a quarter of the set-up's wall-clock."""
    __slots__ = ('_text', '_off', '_a', '_b', '_bytes')

    def __init__(self, text, off, a, b, as_bytes=False):
        self._text, self._off, self._a, self._b, self._bytes = text, off, a, b, as_bytes

    def __len__(self):
        return self._b - self._a

    def _one(self, r):
        piece = self._text[self._off[r]:self._off[r + 1]]
        return piece.decode('utf-8') if self._bytes else piece   # (a blob with non-ASCII names is kept as bytes: the offsets are bytes)

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self._one(self._a + r) for r in range(*k.indices(self._b - self._a))]
        n = self._b - self._a
        if k < 0:
            k += n
        if not 0 <= k < n:
            raise IndexError(k)
        return self._one(self._a + k)

    def __iter__(self):
        return (self._one(r) for r in range(self._a, self._b))

    def __eq__(self, other):
        return list(self) == list(other)


class NativeOverview:
    """overview.csv of one locus, parsed by the library: the `saved` rows' columns as arrays, and later the writer of the locus's
    step-3 files.  open() returns None when the table has to go through pandas (reason in `NativeOverview.last_refusal`)."""
    last_refusal = ''

    def __init__(self, handle, path, owner=None):
        self._h, self.path, self._owner = handle, path, owner   # owner: a NativeSetup that holds (and frees) the handle
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

    @classmethod
    def _from_columns(cls, handle, path, owner, n_rows, saved, reverse, lo, hi, names, run_id, fast5_path):
        """The same object from columns a chunk of loci handed over together (NativeSetup.run_many: wsh_loci_columns)."""
        self = cls.__new__(cls)
        self._h, self.path, self._owner = handle, path, owner
        self.n_saved, self.n_rows = len(saved), n_rows
        self.saved, self.reverse, self.lo, self.hi = saved, reverse, lo, hi
        self._names, self.run_id, self.fast5_path = names, run_id, fast5_path
        return self

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
            if handle:
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
        rc = h.wsh_locus_store(self._h, os.fsencode(locus_path), *[a.ctypes.data for a in arrs], 3 if write else 0)
        if rc != 0:
            raise OSError((h.wsh_locus_error(self._h) or b'').decode('utf-8', 'replace'))
        return self.table_text()

    def table_text(self) -> str:
        """The table the last store made (the text of the new overview.csv)."""
        n = C.c_int64()
        p = lib().wsh_locus_table(self._h, C.byref(n))
        return C.string_at(p, n.value).decode('utf-8')

    def close(self):
        if self._h is not None and self._owner is None and lib() is not None:
            lib().wsh_locus_free(self._h)
        self._h = self._owner = None

    __del__ = close


_COLLAPSE_CONST: dict = {}


def collapse_store(locus_path: str, seq2, off2, len2, reverse, repeat_units, offsets, header: str, sel, write: bool):
    """collapse_repeats for every called sequence + the complex-unit table: (counts [n x alternatives], CSV text), or None when
    the Python form has to run (library absent, an empty alternative)."""
    h = lib()
    if h is None:
        return None
    n = len(len2)
    # (what depends on the pattern alone is made once per pattern: a run's loci repeat a few hundred patterns at most, and this
    # function runs under the interpreter's lock for every locus with more than one unit)
    key = (header, tuple(tuple(u) for u in repeat_units), tuple(int(x) for x in offsets), tuple(int(x) for x in sel))
    const = _COLLAPSE_CONST.get(key)
    if const is None:
        n_alt = np.array([len(u) for u in repeat_units], np.int32)
        flat = [a.encode('ascii') for u in repeat_units for a in u]
        alt_off = np.zeros(len(flat) + 1, np.int32)
        np.cumsum([len(a) for a in flat], out=alt_off[1:])
        if len(_COLLAPSE_CONST) > 4096:
            _COLLAPSE_CONST.clear()
        const = _COLLAPSE_CONST[key] = (n_alt, b''.join(flat), alt_off, np.ascontiguousarray(offsets, np.int32),
                                        np.ascontiguousarray(sel, np.int32), header.encode('utf-8'), len(flat))
    n_alt, blob, alt_off, offs, sel, header_b, n_flat = const
    counts = np.zeros((n, n_flat), np.int64)
    arrs = [np.ascontiguousarray(seq2, np.uint8), np.ascontiguousarray(off2, np.int64), np.ascontiguousarray(len2, np.int32),
            np.ascontiguousarray(reverse, np.uint8)]
    out, ln = C.c_void_p(), C.c_int64()
    rc = h.wsh_collapse_store(os.fsencode(locus_path), n, *[a.ctypes.data for a in arrs], len(repeat_units), n_alt.ctypes.data, blob,
                              alt_off.ctypes.data, offs.ctypes.data, header_b, len(sel), sel.ctypes.data, 1 if write else 0,
                              counts.ctypes.data,
                              C.byref(out), C.byref(ln))
    if rc > 0:
        return None
    if rc < 0:
        raise OSError(f'cannot write the complex-unit table under {locus_path}')
    try:
        return counts, C.string_at(out, ln.value).decode('utf-8')
    finally:
        h.wsh_free(out)


_LAZY_CLASS = None


def _native_table(a: WshAutomaton, kmersize: int, owner):
    """An AutomatonTable whose arrays stay in the library's memory until somebody looks at them: the handle of a run takes
    the pointers as they are (caller.HipCaller), a run of thousands of loci never builds the 14 000 NumPy arrays."""
    global _LAZY_CLASS
    if _LAZY_CLASS is None:   # (made once, and not at import: automata.py imports this module)
        from .automata import AutomatonTable

        class _Lazy(AutomatonTable):
            def __init__(self, a, kmersize, owner):  # noqa: D107 -- no arrays yet
                # a: a WshAutomaton, or its fields as plain numbers (n_states, endstate, repstart, repend, n_edges, then the seven
                # pointers: a chunk of loci reads all its structs at once -- NativeSetup.run_many -- a ctypes field costs ten list items)
                if type(a) is not tuple:
                    a = (a.n_states, a.endstate, a.repstart, a.repend, a.n_edges, a.value, a.seq_idx, a.pred_ptr, a.pred_idx, a.repeat_mask,
                         a.last_base, a.kmer)
                self.n_states, self.endstate, self.repstart, self.repend, self._n_edges = a[:5]
                self.kmersize, self._kmers, self._succ = kmersize, None, None
                self._owner = owner
                self.native_ptrs = a[5:11]
                self._kmer_ptr = a[11]

            def __getattr__(self, name):  # only reached for attributes that are not set yet: the arrays
                if name in ('value', 'seq_idx', 'pred_ptr', 'pred_idx', 'repeat_mask', 'last_base', 'kmer_codes'):
                    S, E = self.n_states, self._n_edges
                    p = self.native_ptrs
                    self.value, self.seq_idx = _copy(p[0], np.float64, S), _copy(p[1], np.int32, S)
                    self.pred_ptr, self.pred_idx = _copy(p[2], np.int32, S + 1), _copy(p[3], np.int32, E)
                    self.repeat_mask, self.last_base = _copy(p[4], np.uint8, S), _copy(p[5], np.uint8, S)
                    self.kmer_codes = _copy(self._kmer_ptr, np.uint32, S)
                    return self.__dict__[name]
                raise AttributeError(name)

            def __deepcopy__(self, memo):
                return AutomatonTable(self.n_states, self.endstate, self.value.copy(), self.seq_idx.copy(), self.pred_ptr.copy(),
                                      self.pred_idx.copy(), self.repeat_mask.copy(), self.last_base.copy(), repstart=self.repstart,
                                      repend=self.repend, kmer_codes=self.kmer_codes.copy(), kmersize=self.kmersize)
        _LAZY_CLASS = _Lazy
    return _LAZY_CLASS(a, kmersize, owner)


def store_many(overviews, locus_paths, start, len1, len2, cost1, cost2, seq2, off2, write: bool):
    """NativeOverview.store for a chunk of loci in ONE library call (no GIL for the whole chunk): locus i's reads are
    [start[i], start[i + 1]) of the run's contiguous per-read arrays (int32 lengths, float64 costs, int64 offsets into seq2)."""
    h = lib()
    n = len(overviews)
    handles = (C.c_void_p * n)(*[o._h for o in overviews])
    paths = (C.c_char_p * n)(*[os.fsencode(p) for p in locus_paths])
    start = np.ascontiguousarray(start, np.int64)
    status = np.zeros(n, np.int32)
    for a, dt in ((len1, np.int32), (len2, np.int32), (cost1, np.float64), (cost2, np.float64), (seq2, np.uint8), (off2, np.int64)):
        assert a.dtype == dt and a.flags.c_contiguous
    bad = h.wsh_loci_store(n, handles, paths, start.ctypes.data, len1.ctypes.data, len2.ctypes.data, cost1.ctypes.data, cost2.ctypes.data,
                           seq2.ctypes.data, off2.ctypes.data, 3 if write else 0, status.ctypes.data)
    if bad:
        i = int(np.flatnonzero(status)[0])
        raise OSError((h.wsh_locus_error(overviews[i]._h) or b'').decode('utf-8', 'replace'))


class NativeSetup:
    """Everything of a locus that precedes the calling, from one library call without the GIL: `overview` (NativeOverview or
    None: the pandas path), `tables` ((template, reverse) AutomatonTables or None: the Python compiler, which raises what
    upstream raises), `similarity` ((CSV text, [warning lines]) or None: the Python form).  None altogether without the library."""

    def __init__(self, raw: WshSetup, locus_path: str, kmersize: int, columns=None, flat=None):
        """flat: the struct's fields as plain numbers (run_many reads a chunk's structs at once): (owner, locus, overview status,
        automata status, similarity status, the two automata's twelve fields each, address of the similarity text, of the warnings)."""
        self._raw = raw
        self.overview = self.tables = self.similarity = None
        if flat is None:
            flat = (raw.owner, raw.locus, raw.overview_status, raw.automata_status, raw.similarity_status, raw.aut[0], raw.aut[1],
                    raw.similarity_csv or b'', raw.warnings or b'')
        locus, ov_status = flat[1], flat[2]
        self.overview_status = ov_status
        if ov_status == 0 and columns is not None:
            self.overview = NativeOverview._from_columns(locus, _under(locus_path, 'overview.csv'), self, *columns)
        elif ov_status == 0:
            self.overview = NativeOverview(locus, _under(locus_path, 'overview.csv'), owner=self)
        else:
            NativeOverview.last_refusal = (lib().wsh_locus_error(locus) or b'').decode('utf-8', 'replace')
        if flat[3] == 0:
            self.tables = (_native_table(flat[5], kmersize, self), _native_table(flat[6], kmersize, self))
        if flat[4] == 0:
            text, warn = flat[7], flat[8]
            if type(text) is int:   # (addresses: C strings in the library's memory)
                text, warn = C.string_at(text) if text else b'', C.string_at(warn) if warn else b''
            self.similarity = (text.decode('ascii'), warn.decode('ascii').splitlines())

    @classmethod
    def run(cls, locus_path: str, sequence: str, pore_model, min_state_similarity: float, write_similarity: bool):
        h = lib()
        if h is None:
            return None
        try:
            seq = sequence.encode('ascii')
        except UnicodeEncodeError:
            return None
        levels = pore_model.level_norm
        if levels.dtype != np.float64 or not levels.flags.c_contiguous:
            levels = np.ascontiguousarray(levels, np.float64)
        raw = WshSetup()
        h.wsh_locus_setup(os.fsencode(locus_path), seq, levels.ctypes.data, int(pore_model.kmersize), float(min_state_similarity),
                          1 if write_similarity else 0, C.byref(raw))
        return cls(raw, locus_path, int(pore_model.kmersize))

    @classmethod
    def run_many(cls, locus_paths, sequences, pore_model, min_state_similarity: float, write_similarity: bool):
        """run() for a chunk of loci in ONE library call (no GIL for the whole chunk); None without the library."""
        h = lib()
        if h is None:
            return None
        try:
            seqs = [s.encode('ascii') for s in sequences]
        except UnicodeEncodeError:
            return None
        n = len(seqs)
        levels = pore_model.level_norm
        if levels.dtype != np.float64 or not levels.flags.c_contiguous:
            levels = np.ascontiguousarray(levels, np.float64)
        raws = (WshSetup * n)()
        h.wsh_loci_setup(n, (C.c_char_p * n)(*[os.fsencode(p) for p in locus_paths]), (C.c_char_p * n)(*seqs), levels.ctypes.data,
                         int(pore_model.kmersize), float(min_state_similarity), 1 if write_similarity else 0, raws)
        # the chunk's structs read at once (a NumPy view of the array, its columns as lists): per locus this loop runs under the
        # interpreter's lock, and a ctypes field costs what ten list items cost
        rec = np.frombuffer(raws, _SETUP_DTYPE, n)
        head = [rec[k].tolist() for k in ('owner', 'locus', 'overview_status', 'automata_status', 'similarity_status', 'similarity_csv', 'warnings')]
        aut = rec['aut']
        ints = np.stack([aut[k] for k in ('n_states', 'endstate', 'repstart', 'repend', 'n_edges')], axis=2).tolist()     # [locus][strand][5]
        ptrs = np.stack([aut[k] for k in ('value', 'seq_idx', 'pred_ptr', 'pred_idx', 'repeat_mask', 'last_base', 'kmer')], axis=2).tolist()
        columns = cls._chunk_columns(h, raws, n, head[2], head[1])
        kmersize = int(pore_model.kmersize)
        out = []
        for i, p in enumerate(locus_paths):   # (each entry keeps the array alive through its element: wsh_setup_free takes the element)
            flat = (head[0][i], head[1][i], head[2][i], head[3][i], head[4][i], tuple(ints[i][0] + ptrs[i][0]), tuple(ints[i][1] + ptrs[i][1]),
                    head[5][i], head[6][i])
            out.append(cls(raws[i], p, kmersize, columns.get(i), flat))
        return out

    @staticmethod
    def _chunk_columns(h, raws, n, status=None, locus=None):
        """{locus of the chunk: (rows, saved rows, reverse, lo, hi, names, run ids or None, fast5 paths or None)} for the loci whose
        overview the library parsed -- two library calls and a handful of arrays for the chunk instead of one call and seven
        copies per locus (this part runs under the interpreter's lock: it, not the parsing, set the set-up's wall-clock)."""
        if status is None:
            status, locus = [raws[i].overview_status for i in range(n)], [raws[i].locus for i in range(n)]
        idx = [i for i in range(n) if status[i] == 0]
        if not idx or not hasattr(h, 'wsh_loci_columns'):
            return {}
        m = len(idx)
        handles = (C.c_void_p * m)(*[locus[i] for i in idx])
        counts, rows, flags, sbytes = np.zeros(m, np.int64), np.zeros(m, np.int64), np.zeros(m, np.int32), np.zeros(3, np.int64)
        h.wsh_loci_counts(m, handles, counts.ctypes.data, rows.ctypes.data, flags.ctypes.data, sbytes.ctypes.data)
        total = int(counts.sum())
        saved, lo, hi = np.zeros(total, np.int64), np.zeros(total, np.int64), np.zeros(total, np.int64)
        reverse = np.zeros(total, np.uint8)
        blobs = [C.create_string_buffer(max(int(b), 1)) for b in sbytes]
        offs = [np.zeros(total + 1, np.int64) for _ in range(3)]
        h.wsh_loci_columns(m, handles, saved.ctypes.data, reverse.ctypes.data, lo.ctypes.data, hi.ctypes.data, blobs[0], offs[0].ctypes.data,
                           blobs[1], offs[1].ctypes.data, blobs[2], offs[2].ctypes.data)
        reverse = reverse.astype(bool)

        def strings(k):   # (text, offsets, the text is bytes) of column k: _Strings cuts a locus's rows out of it when asked
            raw, o = blobs[k].raw[:int(sbytes[k])], offs[k].tolist()
            text = raw.decode('utf-8')
            return (text, o, False) if len(text) == len(raw) else (raw, o, True)   # (non-ASCII: the offsets are bytes)
        names = strings(0)
        runs = strings(1) if (flags & 1).any() else None
        f5s = strings(2) if (flags & 2).any() else None
        out, at = {}, 0
        counts_l, rows_l, flags_l = counts.tolist(), rows.tolist(), flags.tolist()
        for q, i in enumerate(idx):
            b = at + counts_l[q]
            out[i] = (rows_l[q], saved[at:b].copy(), reverse[at:b].copy(), lo[at:b].copy(), hi[at:b].copy(), _Strings(names[0], names[1], at, b, names[2]),
                      _Strings(runs[0], runs[1], at, b, runs[2]) if flags_l[q] & 1 else None,
                      _Strings(f5s[0], f5s[1], at, b, f5s[2]) if flags_l[q] & 2 else None)
            at = b
        return out

    def __del__(self):
        h = lib()
        if h is not None and getattr(self, '_raw', None) is not None and self._raw.owner:
            h.wsh_setup_free(C.byref(self._raw))
