"""Host side of the HIP caller: the reference's step-3 interface over the C ABI.

Mirrors (same names, argument meaning and result fields):
  ReadSignal                    src/schemas/readsignal.py:6-10
  CallerResult                  src/caller/caller.py:46-51
  CallerConfig / RescalerConfig src/config.py:91-119
  CallerWrapper.run(workload)   src/caller/wrapper.py:104-120  (order-preserving map over reads)
  WarpSTR.warp                  src/caller/caller.py:189-193   (HipCaller.warp)
All per-read arithmetic runs on the GPU (libwarpstr_hip.so); this module only packs buffers, and
turns state paths into base strings (WarpSTR._get_sequence, src/caller/caller.py:178-187).
"""
import collections.abc
import ctypes as C
import os
import threading
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from .automata import AutomatonTable, locus_automata


@dataclass
class ReadSignal:
    name: str
    reverse: bool
    signal: np.ndarray


@dataclass
class CallerResult:
    seq: str
    cost: float
    resc_seq: str
    resc_cost: float


@dataclass
class RescalerConfig:
    reps_as_one: bool = False
    threshold: float = 0.5
    max_std: float = 0.5
    method: str = 'mean'

    def __post_init__(self):
        assert self.threshold > 0
        assert self.max_std > 0
        assert self.method == 'mean' or self.method == 'median'


@dataclass
class CallerConfig:
    spike_removal: str = 'Brute'
    min_values_per_state: int = 4
    states_in_segment: int = 6
    min_state_similarity: float = 0.75
    visualize_alignment: bool = True
    visualize_phase: bool = True
    visualize_strand: bool = True
    visualize_cost: bool = True

    def __post_init__(self):
        assert self.min_values_per_state > 1
        assert self.states_in_segment > 1
        assert self.min_state_similarity > 0
        assert self.spike_removal in ['None', 'median3', 'median5', 'Brute']


class ReadCallError(RuntimeError):
    """A read could not be called (the reference would have raised inside its worker)."""


_COMPLEMENT = str.maketrans('ACGT', 'TGCA')


def pack_signals(signals: Sequence[np.ndarray]):
    """Concatenate squiggles -> (float64 buffer, int64 offsets[n+1])."""
    lens = np.fromiter((len(s) for s in signals), dtype=np.int64, count=len(signals))
    offsets = np.zeros(len(signals) + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    if len(signals) == 0:
        return np.empty(0, dtype=np.float64), offsets
    buf = np.concatenate(signals)  # one pass in C (a Python loop of slice assignments cost 78 ms per 50k reads)
    if buf.dtype != np.float64:
        buf = buf.astype(np.float64)
    return buf, offsets


class CallerResults(collections.abc.Sequence):
    """What CallerWrapper.run returns: a list-like of CallerResult (src/caller/caller.py:46-51) in workload order, backed by
    the batch's result records and the two ASCII sequence buffers -- a CallerResult (two Python strings) is only built for
    the reads that are looked at.  The records are also available as columns (`len2`, `cost2`, ...) for batch consumers."""

    def __init__(self, names, records, offsets, seq1, seq2, on_error: str, offsets2=None):
        """seq1 / seq2: the batch's ASCII buffers (bytes or uint8 arrays; read r's seq starts at offsets[r], its resc_seq at
        offsets2[r] -- the same place in the library's per-sample layout, different ones once packed)."""
        self.names, self.records, self.offsets = names, records, np.asarray(offsets)
        self.offsets2 = self.offsets if offsets2 is None else np.asarray(offsets2)
        self._seq1, self._seq2, self._on_error = seq1, seq2, on_error

    def __len__(self):
        return len(self.records)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        rec = self.records[i]
        st = int(rec['status'])
        if st != 0:
            if self._on_error == 'raise':
                raise ReadCallError(f'read {self.names[i]}: caller status {_lib.READ_STATUS.get(st, st)}')
            return CallerResult('', float('nan'), '', float('nan'))
        o, o2 = int(self.offsets[i]), int(self.offsets2[i])
        return CallerResult(seq=bytes(self._seq1[o:o + int(rec['len1'])]).decode('ascii'), cost=float(rec['cost1']),
                            resc_seq=bytes(self._seq2[o2:o2 + int(rec['len2'])]).decode('ascii'), resc_cost=float(rec['cost2']))

    def check(self):
        """Raise for the first read the reference would have failed on (on_error='raise'); cheap, vectorised."""
        bad = np.flatnonzero(self.records['status'] != 0)
        if len(bad) and self._on_error == 'raise':
            st = int(self.records['status'][bad[0]])
            raise ReadCallError(f'read {self.names[bad[0]]}: caller status {_lib.READ_STATUS.get(st, st)}')
        return self

    def sequences(self):
        """([seq ...], [resc_seq ...]) of all reads at once ('' for a read that was not called, as __getitem__ gives with
        on_error='nan'; call check() first to raise instead): one decode of each buffer instead of two objects per read."""
        ok = self.records['status'] == 0
        out = []
        for buf, offs, field in ((self._seq1, self.offsets, 'len1'), (self._seq2, self.offsets2, 'len2')):
            text = (buf.tobytes() if isinstance(buf, np.ndarray) else bytes(buf)).decode('ascii', 'replace')
            lens = np.where(ok, self.records[field], 0)
            out.append([text[o:o + n] for o, n in zip(np.asarray(offs).tolist(), lens.tolist())])
        return out[0], out[1]

    def lengths(self):
        """(len(seq), len(resc_seq)) per read as two integer arrays: all that overview.csv stores of the sequences."""
        return self.records['len1'].copy(), self.records['len2'].copy()

    def costs(self):
        return self.records['cost1'].copy(), self.records['cost2'].copy()


class _WorkloadNames(collections.abc.Sequence):
    """names[i] = workload[i].name without building a 50 000-element list per call."""

    def __init__(self, workload):
        self._w = workload

    def __len__(self):
        return len(self._w)

    def __getitem__(self, i):
        return self._w[i].name


_SEAM = False


def _seam():
    """warpstr_amd/_seam_helper.so (csrc/seam_helper.c: the two per-read loops of the Python seam in C), or None."""
    global _SEAM
    if _SEAM is False:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_seam_helper.so')
        _SEAM = None
        if os.path.exists(path) and not os.environ.get('WARPSTR_NO_SEAM_HELPER'):
            try:
                lib = C.PyDLL(path)
                lib.wsx_seam_collect.restype = C.c_int64
                lib.wsx_seam_collect.argtypes = [C.py_object, C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.py_object]
                lib.wsx_seam_buffers_i16.restype = C.c_int64
                lib.wsx_seam_buffers_i16.argtypes = [C.py_object, C.c_void_p, C.c_void_p]
                lib.wsx_seam_pack_sequences.restype = C.c_int64
                lib.wsx_seam_pack_sequences.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                                        C.c_void_p]
                _SEAM = lib
            except OSError:
                pass
    return _SEAM


# caller_config.spike_removal (src/config.py, remove_spikes src/schemas/fast5.py:68-75) -> wsx_prepare_signals' code
SPIKE_REMOVAL = {'None': 0, 'Brute': 1, 'median3': 2, 'median5': 3}


class HipCaller:
    """One handle on one GPU holding the automata of one or more loci (include/warpstr_hip.h)."""

    # launch-policy knobs applied to every new handle (names: _lib.TUNING); the GPU tests set it to force fallback kernels
    default_tuning: dict = {}

    def __init__(self, automata: Sequence[AutomatonTable], flank_lengths: Sequence[int],
                 caller_config: Optional[CallerConfig] = None, rescaler_config: Optional[RescalerConfig] = None,
                 device: int = 0, stream: int = 0, workspace_limit: Optional[int] = None,
                 reverse_flags: Optional[Sequence[bool]] = None):
        """reverse_flags[i]: automaton i belongs to the reverse strand (its called sequences are reverse-complemented);
        default: odd positions (template, reverse, template, reverse, ...)."""
        import time
        t_init = time.perf_counter()
        self.lib = _lib.load()
        self.device = int(device)
        if self.lib.wsx_device_count() <= 0:
            raise RuntimeError('warpstr_amd: no HIP device visible; the caller has no CPU path')
        self.caller_config = caller_config or CallerConfig()
        self.rescaler_config = rescaler_config or RescalerConfig()
        self.automata, self.flank_lengths, self.reverse_flags = [], [], []
        arr, keep = self._pack(automata, flank_lengths, reverse_flags)
        prm = _lib.WsxParams(self.caller_config.min_values_per_state, self.caller_config.states_in_segment,
                             self.rescaler_config.threshold, self.rescaler_config.max_std,
                             1 if self.rescaler_config.method == 'median' else 0,
                             1 if self.rescaler_config.reps_as_one else 0)
        self.handle = C.c_void_p()
        self._seq_landing = None
        t_call = time.perf_counter()
        _lib.check(self.lib.wsx_caller_create(C.byref(self.handle), device, C.byref(arr), len(self.automata),
                                              C.byref(prm), C.c_void_p(stream)), 'wsx_caller_create')
        del keep   # (the library has copied what it needs)
        self.init_s = {'host_tables': t_call - t_init, 'wsx_caller_create': time.perf_counter() - t_call}
        if workspace_limit:
            _lib.check(self.lib.wsx_caller_set_workspace_limit(self.handle, workspace_limit),
                       'wsx_caller_set_workspace_limit')
        self.max_states = max(t.n_states for t in self.automata)
        for knob, value in self.default_tuning.items():
            self.set_tuning(knob, value)

    def _pack(self, automata, flank_lengths, reverse_flags):
        """The wsx_automaton array of `automata` (and what keeps its pointers valid until the library has copied the tables);
        the handle's lists grow by them.  reverse_flags default: odd positions OF THIS CALL are reverse-strand automata."""
        automata = list(automata)
        flank_lengths = [int(f) for f in flank_lengths]
        if reverse_flags is None:
            reverse_flags = [bool(i & 1) for i in range(len(automata))]
        reverse_flags = [bool(x) for x in reverse_flags]
        if not (len(automata) == len(flank_lengths) == len(reverse_flags)):
            raise ValueError('one flank length and one strand flag per automaton')
        arr = (_lib.WsxAutomaton * len(automata))()
        keep = []
        for i, (t, fl) in enumerate(zip(automata, flank_lengths)):
            ptrs = t.__dict__.get('native_ptrs') if 'value' not in t.__dict__ else None
            if ptrs is not None:  # tables still in the host library's memory (_hostlib.NativeSetup): the pointers as they are
                keep.append(t)
                arr[i] = _lib.WsxAutomaton(t.n_states, t.endstate, fl, 1 if reverse_flags[i] else 0, *ptrs)
                continue
            bufs = [np.ascontiguousarray(t.value, np.float64), np.ascontiguousarray(t.seq_idx, np.int32),
                    np.ascontiguousarray(t.pred_ptr, np.int32), np.ascontiguousarray(t.pred_idx, np.int32),
                    np.ascontiguousarray(t.repeat_mask, np.uint8), np.ascontiguousarray(t.last_base, np.uint8)]
            keep.append(bufs)
            arr[i] = _lib.WsxAutomaton(t.n_states, t.endstate, fl, 1 if reverse_flags[i] else 0, *[_lib.ptr(b) for b in bufs])
        self.automata += automata
        self.flank_lengths += flank_lengths
        self.reverse_flags += reverse_flags
        return arr, keep

    def add_automata(self, automata: Sequence[AutomatonTable], flank_lengths: Sequence[int],
                     reverse_flags: Optional[Sequence[bool]] = None) -> int:
        """wsx_caller_add_automata: more automata for a handle that is in use (the loci of the next group of a run, while the
        reads of the last are on the device) -> the index of the first of them.  Earlier automata keep their indices; calls
        already enqueued are not disturbed.  From the thread that makes the handle's calls."""
        import time
        t0 = time.perf_counter()
        n_before = len(self.automata)
        arr, keep = self._pack(automata, flank_lengths, reverse_flags)
        first = C.c_int32(-1)
        t1 = time.perf_counter()
        rc = self.lib.wsx_caller_add_automata(self.handle, C.byref(arr), len(arr), C.byref(first))
        if rc != 0:   # (the handle is as it was)
            del self.automata[n_before:], self.flank_lengths[n_before:], self.reverse_flags[n_before:]
        _lib.check(rc, 'wsx_caller_add_automata')
        del keep
        self.max_states = max(self.max_states, max(t.n_states for t in self.automata[n_before:]))
        self.init_s['host_tables'] += t1 - t0
        self.init_s['wsx_caller_add_automata'] = self.init_s.get('wsx_caller_add_automata', 0.0) + time.perf_counter() - t1
        return int(first.value)

    def create_times(self) -> dict:
        """Where wsx_caller_create spent its time (seconds): a handle for all loci of a run places thousands of automata."""
        v = (C.c_double * 5)()
        _lib.check(self.lib.wsx_caller_create_times(self.handle, v, 5), 'wsx_caller_create_times')
        return dict(zip(('validate', 'placement', 'pack', 'upload', 'streams'), [float(x) for x in v]), **{f'host_{k}': v for k, v in self.init_s.items()})

    def set_tuning(self, knob: str, value: int):
        """wsx_caller_set_tuning: how the work is spread over launches, never what is computed (knobs: _lib.TUNING)."""
        _lib.check(self.lib.wsx_caller_set_tuning(self.handle, _lib.TUNING[knob], int(value)), 'wsx_caller_set_tuning')

    def workspace_limit(self) -> int:
        v = C.c_uint64()
        _lib.check(self.lib.wsx_caller_get_workspace_limit(self.handle, C.byref(v)), 'wsx_caller_get_workspace_limit')
        return int(v.value)

    def close(self):
        if getattr(self, 'handle', None) is not None and self.handle:
            self.lib.wsx_caller_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def kernel_name(self, automaton: int = 0) -> str:
        return self.lib.wsx_caller_kernel_name(self.handle, automaton).decode()

    # ---- host-buffer entry points -------------------------------------------------------------
    def call(self, signal: np.ndarray, offsets: np.ndarray, automaton_id: np.ndarray, want_traces: bool = False,
             want_debug: bool = False, want_seqs: bool = False):
        """wsx_call_batch on host buffers -> (results structured array, dict of optional per-sample outputs)."""
        signal = np.ascontiguousarray(signal, np.float64)
        offsets = np.ascontiguousarray(offsets, np.int64)
        automaton_id = np.ascontiguousarray(automaton_id, np.int32)
        n = len(automaton_id)
        assert len(offsets) == n + 1 and offsets[0] == 0 and offsets[-1] == len(signal)
        results = np.zeros(n, dtype=_lib.RESULT_DTYPE)
        extra = {}
        tr = None
        if want_traces or want_debug or want_seqs:
            if want_traces or want_debug:
                extra['trace1'] = np.zeros(len(signal), np.uint16)
                extra['trace2'] = np.zeros(len(signal), np.uint16)
            if want_debug:
                extra['rescaled'] = np.zeros(len(signal), np.float64)
                extra['badmask'] = np.zeros(len(signal), np.uint8)
            if want_seqs:
                extra['seq1'] = np.zeros(len(signal), np.uint8)
                extra['seq2'] = np.zeros(len(signal), np.uint8)
            tr = _lib.WsxTraces(_lib.ptr(extra.get('trace1')), _lib.ptr(extra.get('trace2')), _lib.ptr(extra.get('rescaled')),
                                _lib.ptr(extra.get('badmask')), _lib.ptr(extra.get('seq1')), _lib.ptr(extra.get('seq2')))
        _lib.check(self.lib.wsx_call_batch(self.handle, _lib.WSX_MEM_HOST, _lib.ptr(signal), _lib.ptr(offsets),
                                           _lib.ptr(automaton_id), n, _lib.ptr(results),
                                           C.byref(tr) if tr is not None else None), 'wsx_call_batch')
        return results, extra

    def call_workload(self, workload: Sequence, want_seqs: bool = True):
        """call_reads for a workload of objects with `.signal` and `.reverse` (ReadSignal, src/schemas/readsignal.py:6-10):
        the pointers and the strand flags are collected by csrc/seam_helper.c when that helper is built (22 ms -> 3 ms
        per 50 000 reads); the automaton of a read is 1 for reverse-strand reads, else 0 (wrapper.py:115)."""
        n = len(workload)
        seam = _seam()
        if seam is not None:
            ptrs, lens, aut = np.empty(n, np.uintp), np.empty(n, np.int64), np.empty(n, np.int32)
            keep = []  # the signal objects themselves: alive until the call has returned
            rc = seam.wsx_seam_collect(workload, b'signal', b'reverse', _lib.ptr(ptrs), _lib.ptr(lens), _lib.ptr(aut), keep)
            if rc == n:
                return self._call_read_ptrs(ptrs, lens, aut, want_seqs, keep=keep)
        aut = np.fromiter((1 if w.reverse else 0 for w in workload), dtype=np.int32, count=n)
        return self.call_reads([np.asarray(w.signal) for w in workload], aut, want_seqs)

    def call_reads(self, reads: Sequence[np.ndarray], automaton_id: np.ndarray, want_seqs: bool = False):
        """wsx_call_batch_reads: reads in separate float64 arrays (a workload of ReadSignal objects), gathered by the library
        while it uploads -> (results, offsets, dict of optional per-sample outputs laid out by `offsets`)."""
        n = len(reads)
        ptrs = np.empty(n, np.uintp)
        lens = np.empty(n, np.int64)
        keep = []
        for i, a in enumerate(reads):
            if a.dtype != np.float64 or a.ndim != 1 or not a.flags.c_contiguous:
                a = np.ascontiguousarray(a, np.float64).reshape(-1)
                keep.append(a)
            ptrs[i] = a.__array_interface__['data'][0]
            lens[i] = a.shape[0]
        return self._call_read_ptrs(ptrs, lens, automaton_id, want_seqs, keep)

    def _call_read_ptrs(self, ptrs, lens, automaton_id, want_seqs, keep):
        n = len(ptrs)
        automaton_id = np.ascontiguousarray(automaton_id, np.int32)
        offsets = np.zeros(n + 1, np.int64)
        np.cumsum(lens, out=offsets[1:])
        results = np.zeros(n, dtype=_lib.RESULT_DTYPE)
        extra, tr = {}, None
        seam = _seam() if want_seqs else None
        if want_seqs:
            total = int(offsets[-1])
            if seam is not None:
                # per-sample landing buffers that live with the handle (their pages are touched once, not per call); what
                # the caller gets are packed copies: sum(len) bytes instead of one byte per sample
                if self._seq_landing is None or len(self._seq_landing[0]) < total:
                    self._seq_landing = (np.empty(total, np.uint8), np.empty(total, np.uint8))
                s1, s2 = self._seq_landing
            else:
                s1, s2 = np.zeros(total, np.uint8), np.zeros(total, np.uint8)
            tr = _lib.WsxTraces(None, None, None, None, _lib.ptr(s1), _lib.ptr(s2))
        _lib.check(self.lib.wsx_call_batch_reads(self.handle, _lib.ptr(ptrs), _lib.ptr(lens), _lib.ptr(automaton_id), n,
                                                 _lib.ptr(results), C.byref(tr) if tr is not None else None),
                   'wsx_call_batch_reads')
        del keep
        if want_seqs and seam is not None:
            for key, src, field in (('seq1', s1, 'len1'), ('seq2', s2, 'len2')):
                lens_f = np.where(results['status'] == 0, results[field], 0).astype(np.int32)  # failed reads: no sequence
                pos = np.empty(n + 1, np.int64)
                out = np.empty(int(lens_f.sum(dtype=np.int64)), np.uint8)
                seam.wsx_seam_pack_sequences(_lib.ptr(src), _lib.ptr(offsets), _lib.ptr(lens_f), 4, n, _lib.ptr(out),
                                             _lib.ptr(pos))
                extra[key], extra[key + '_pos'] = out, pos
        elif want_seqs:
            extra['seq1'], extra['seq2'] = s1, s2
            extra['seq1_pos'] = extra['seq2_pos'] = offsets
        return results, offsets, extra

    def warp(self, signal: np.ndarray, offsets: np.ndarray, automaton_id: np.ndarray, mask: Optional[np.ndarray] = None,
             want_last_row: bool = False):
        """wsx_warp_batch on host buffers -> dict(trace, end_cost, status[, last_row])."""
        signal = np.ascontiguousarray(signal, np.float64)
        offsets = np.ascontiguousarray(offsets, np.int64)
        automaton_id = np.ascontiguousarray(automaton_id, np.int32)
        n = len(automaton_id)
        trace = np.zeros(len(signal), np.uint16)
        end_cost = np.zeros(n, np.float64)
        status = np.zeros(n, np.int32)
        stride = self.max_states
        last_row = np.full((n, stride), np.nan) if want_last_row else None
        mk = np.ascontiguousarray(mask, np.uint8) if mask is not None else None
        _lib.check(self.lib.wsx_warp_batch(self.handle, _lib.WSX_MEM_HOST, _lib.ptr(signal), _lib.ptr(offsets),
                                           _lib.ptr(automaton_id), n, _lib.ptr(mk), _lib.ptr(trace), _lib.ptr(end_cost),
                                           _lib.ptr(last_row), stride, _lib.ptr(status)), 'wsx_warp_batch')
        out = dict(trace=trace, end_cost=end_cost, status=status)
        if want_last_row:
            out['last_row'] = last_row
        return out

    def prepare_signals(self, raws: Sequence[np.ndarray], positions: Sequence[Sequence[int]], spike_removal: str = 'Brute'):
        """wsx_prepare_signals on host buffers: raw int16 reads + (l_start_raw, r_end_raw) ->
        (normalised float64 buffer, offsets) ready for call(); also returns the (shift, scale) table."""
        spike_code = SPIKE_REMOVAL[spike_removal]
        n = len(raws)
        lens = np.fromiter((len(r) for r in raws), dtype=np.int64, count=n)
        roff = np.zeros(n + 1, np.int64)
        np.cumsum(lens, out=roff[1:])
        raw = np.empty(int(roff[-1]), np.int16)
        for r, o in zip(raws, roff[:-1]):
            raw[o:o + len(r)] = r
        lo = np.array([p[0] for p in positions], np.int64)
        hi = np.array([p[1] for p in positions], np.int64)
        seglen = np.array([len(range(*slice(int(a), int(b) + 1).indices(int(L)))) for a, b, L in zip(lo, hi, lens)], np.int64)
        ooff = np.zeros(n + 1, np.int64)
        np.cumsum(seglen, out=ooff[1:])
        out = np.empty(int(ooff[-1]), np.float64)
        ss = np.zeros((n, 2), np.float64)
        _lib.check(self.lib.wsx_prepare_signals(self.handle, _lib.WSX_MEM_HOST, _lib.ptr(raw), _lib.ptr(roff), _lib.ptr(lo),
                                                _lib.ptr(hi), n, spike_code, _lib.ptr(out),
                                                _lib.ptr(ooff), _lib.ptr(ss)), 'wsx_prepare_signals')
        return out, ooff, ss

    # ---- device-buffer entry points (pointers from torch tensors) ------------------------------
    def prepare_device(self, raw_ptr: int, raw_offsets: np.ndarray, seg_start: np.ndarray, seg_end: np.ndarray, out_ptr: int,
                       out_offsets: np.ndarray, spike_removal: str = 'Brute', shift_scale_ptr: int = 0):
        """wsx_prepare_signals on device buffers: int16 raw reads in HBM -> normalised float64 segments in HBM, laid out
        by out_offsets (ready for call_device with the same offsets).  Asynchronous: enqueued on the handle's stream, a
        call_device that follows reads the segments in stream order."""
        raw_offsets = np.ascontiguousarray(raw_offsets, np.int64)
        seg_start, seg_end = np.ascontiguousarray(seg_start, np.int64), np.ascontiguousarray(seg_end, np.int64)
        out_offsets = np.ascontiguousarray(out_offsets, np.int64)
        _lib.check(self.lib.wsx_prepare_signals(self.handle, _lib.WSX_MEM_DEVICE, C.c_void_p(raw_ptr), _lib.ptr(raw_offsets),
                                                _lib.ptr(seg_start), _lib.ptr(seg_end), len(seg_start),
                                                SPIKE_REMOVAL[spike_removal], C.c_void_p(out_ptr), _lib.ptr(out_offsets),
                                                C.c_void_p(shift_scale_ptr or None)), 'wsx_prepare_signals')

    def vbz_decode_device(self, src_ptr: int, src_bytes: int, blocks: np.ndarray, dst_ptr: int, dst_samples: int, status_ptr: int = 0):
        """wsx_vbz_decode: StreamVByte blocks (what is inside a VBZ chunk's zstd frame) or plain int16 samples in HBM -> the
        int16 samples in HBM, where `blocks` (_lib.VBZ_BLOCK_DTYPE) says.  Asynchronous, on the handle's stream: a prepare_device
        that follows reads the samples in stream order."""
        blocks = np.ascontiguousarray(blocks, _lib.VBZ_BLOCK_DTYPE)
        _lib.check(self.lib.wsx_vbz_decode(self.handle, C.c_void_p(src_ptr), int(src_bytes), _lib.ptr(blocks), len(blocks),
                                           C.c_void_p(dst_ptr), int(dst_samples), C.c_void_p(status_ptr or None)), 'wsx_vbz_decode')

    def zstd_decode_device(self, src_ptr: int, src_bytes: int, frames: np.ndarray, dst_ptr: int, dst_bytes: int, scratch_ptr: int,
                           status_ptr: int = 0):
        """wsx_zstd_decode: zstd frames in HBM (the bodies of VBZ chunks as they lie in the file) -> their content in HBM, where
        `frames` (_lib.ZSTD_FRAME_DTYPE) says; scratch: as large as dst.  Asynchronous, on the handle's stream: a
        vbz_decode_device that follows reads the content in stream order."""
        frames = np.ascontiguousarray(frames, _lib.ZSTD_FRAME_DTYPE)
        _lib.check(self.lib.wsx_zstd_decode(self.handle, C.c_void_p(src_ptr), int(src_bytes), _lib.ptr(frames), len(frames), C.c_void_p(dst_ptr),
                                            int(dst_bytes), C.c_void_p(scratch_ptr), C.c_void_p(status_ptr or None)), 'wsx_zstd_decode')

    def call_device(self, signal_ptr: int, offsets: np.ndarray, automaton_id: np.ndarray, results_ptr: int,
                    trace1_ptr: int = 0, trace2_ptr: int = 0, seq1_ptr: int = 0, seq2_ptr: int = 0):
        """wsx_call_batch on device buffers; asynchronous (see set_pipelined / join / synchronize)."""
        offsets = np.ascontiguousarray(offsets, np.int64)
        automaton_id = np.ascontiguousarray(automaton_id, np.int32)
        tr = None
        if trace1_ptr or trace2_ptr or seq1_ptr or seq2_ptr:
            tr = _lib.WsxTraces(C.c_void_p(trace1_ptr or None), C.c_void_p(trace2_ptr or None), None, None,
                                C.c_void_p(seq1_ptr or None), C.c_void_p(seq2_ptr or None))
        _lib.check(self.lib.wsx_call_batch(self.handle, _lib.WSX_MEM_DEVICE, C.c_void_p(signal_ptr), _lib.ptr(offsets),
                                           _lib.ptr(automaton_id), len(automaton_id), C.c_void_p(results_ptr),
                                           C.byref(tr) if tr is not None else None), 'wsx_call_batch')

    def set_streams(self, n: int):
        _lib.check(self.lib.wsx_caller_set_streams(self.handle, n), 'wsx_caller_set_streams')

    def set_pipelined(self, on: bool = True):
        """Device-buffer calls stop joining the handle's stream at their end, so that back-to-back calls overlap
        (include/warpstr_hip.h: wsx_caller_set_pipelined); consume outputs after join() or synchronize()."""
        _lib.check(self.lib.wsx_caller_set_pipelined(self.handle, int(bool(on))), 'wsx_caller_set_pipelined')

    def join(self, stream: int = 0):
        """Order `stream` (a hipStream_t value; 0 = the handle's stream) after every call enqueued so far."""
        _lib.check(self.lib.wsx_caller_join(self.handle, C.c_void_p(stream or None)), 'wsx_caller_join')

    def synchronize(self):
        _lib.check(self.lib.wsx_caller_synchronize(self.handle), 'wsx_caller_synchronize')

    def timing_window(self, on: bool = True):
        """last_timing() covers every call from now on (on) or the most recent call only (off)."""
        _lib.check(self.lib.wsx_caller_timing_window(self.handle, int(bool(on))), 'wsx_caller_timing_window')

    def last_timing(self):
        dp, nl, tot = C.c_double(), C.c_int32(), C.c_double()
        _lib.check(self.lib.wsx_caller_last_timing(self.handle, C.byref(dp), C.byref(nl), C.byref(tot)),
                   'wsx_caller_last_timing')
        return dict(dp_kernel_ms=dp.value, dp_launches=nl.value, total_ms=tot.value)

    def workspace(self):
        """Device memory held by the handle, and per sample of the most recent call (wsx_caller_workspace)."""
        total, per = C.c_uint64(), C.c_double()
        _lib.check(self.lib.wsx_caller_workspace(self.handle, C.byref(total), C.byref(per)), 'wsx_caller_workspace')
        return dict(bytes_allocated=int(total.value), bytes_per_sample=per.value)

    def fill_intervals(self):
        """(begin_ms, end_ms, reads) of every fill launch behind last_timing(), relative to the start of the timed region."""
        n = C.c_int32()
        _lib.check(self.lib.wsx_caller_fill_intervals(self.handle, None, None, None, 0, C.byref(n)), 'wsx_caller_fill_intervals')
        b, e, r = np.zeros(n.value), np.zeros(n.value), np.zeros(n.value, np.int32)
        _lib.check(self.lib.wsx_caller_fill_intervals(self.handle, _lib.ptr(b), _lib.ptr(e), _lib.ptr(r), n.value, C.byref(n)),
                   'wsx_caller_fill_intervals')
        return b, e, r


def ragged_index(starts: np.ndarray, lens: np.ndarray) -> np.ndarray:
    """Element indices of the ragged pieces [starts[k], starts[k] + lens[k]) laid end to end (one int64 per element)."""
    lens = np.asarray(lens, np.int64)
    total = int(lens.sum())
    if total == 0:
        return np.zeros(0, np.int64)
    pos = np.cumsum(lens) - lens
    return np.repeat(np.asarray(starts, np.int64) - pos, lens) + np.arange(total, dtype=np.int64)


def slice_lengths(lo: np.ndarray, hi: np.ndarray, lens: np.ndarray) -> np.ndarray:
    """len(signal[lo : hi + 1]) per read with Python's slice rules (Fast5.get_data_processed, src/schemas/fast5.py:56)."""
    lo, hi, lens = np.asarray(lo, np.int64), np.asarray(hi, np.int64), np.asarray(lens, np.int64)
    if len(lo) and (lo.min() < 0 or hi.min() < -1):  # negative indices count from the end: rare, keep Python's own arithmetic
        return np.array([len(range(*slice(int(a), int(b) + 1).indices(int(L)))) for a, b, L in zip(lo, hi, lens)], np.int64)
    return np.clip(np.minimum(hi + 1, lens) - np.minimum(lo, lens), 0, None)


class SharedStaging:
    """A ring of int16 staging buffers that OTHER PROCESSES can fill: files under /dev/shm mapped here and -- by path -- in the
    reader processes of a many-loci run (loci.py), which decode fast5 reads straight into them; `register(address, bytes)` (the
    GPU side: hipHostRegister through torch) makes a mapping page-locked so that the upload starts from it without another copy.
    A slot is {'path', 'view' (np.int16), 'event' (None or something with .synchronize(): its last upload)}."""

    def __init__(self, n_slots: int = 3, register=None, unregister=None, directory: str = '/dev/shm'):
        if not (os.path.isdir(directory) and os.access(directory, os.W_OK)):
            raise OSError(f'{directory} is not a writable directory')
        self._dir, self._register, self._unregister = directory, register, unregister
        self._slots = [dict(path=None, view=None, event=None, mm=None, registered=False) for _ in range(n_slots)]
        self._turn = 0

    def take(self, count: int) -> dict:
        """The next slot of the ring with room for `count` samples, its previous upload finished."""
        import mmap
        import tempfile
        slot = self._slots[self._turn % len(self._slots)]
        self._turn += 1
        if slot['event'] is not None:
            slot['event'].synchronize()
            slot['event'] = None
        if slot['view'] is None or slot['view'].size < count:
            self._release(slot)
            cap = max(count + count // 4, 1 << 22)
            st = os.statvfs(self._dir)   # (a memory-backed file system takes the truncate and faults on the first write past its limit)
            if st.f_bavail * st.f_frsize < cap * 2 + (64 << 20):
                raise OSError(f'{self._dir} has no room for a staging buffer of {cap * 2} bytes')
            fd, path = tempfile.mkstemp(prefix='warpstr_stage_', dir=self._dir)
            try:
                os.ftruncate(fd, cap * 2)
                mm = mmap.mmap(fd, cap * 2, flags=mmap.MAP_SHARED | getattr(mmap, 'MAP_POPULATE', 0))   # (pages present from the start)
            finally:
                os.close(fd)
            view = np.frombuffer(mm, dtype=np.int16)
            slot.update(path=path, view=view, mm=mm, registered=False)
            if self._register is not None:
                slot['registered'] = bool(self._register(view.ctypes.data, cap * 2))
        return slot

    def _release(self, slot):
        if slot['view'] is not None:
            if slot['registered'] and self._unregister is not None:
                self._unregister(slot['view'].ctypes.data)
            slot['view'] = None
            slot['mm'] = None   # (the mapping goes with its last reference)
            try:
                os.unlink(slot['path'])
            except OSError:
                pass
            slot['path'] = None

    def close(self):
        for slot in self._slots:
            if slot['event'] is not None:
                slot['event'].synchronize()
                slot['event'] = None
            self._release(slot)

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass


class ArenaTable:
    """The reader processes' arenas (memory-backed files, warpstr_amd/_readers.py) as this process uses them: mapped, and registered
    with the runtime so that an upload starts from them directly -- once per file and size, by whichever thread asks first."""

    def __init__(self, device):
        import torch
        self.torch = torch
        self.dev = device if isinstance(device, torch.device) else torch.device('cuda', int(device))
        self.entries = {}    # path -> (mmap, page-locked int16 tensor, registered, address)
        self.lock = threading.Lock()   # guards `entries` and `busy`; page-locking itself runs under the PATH's lock, several at a time
        self.busy = {}       # path -> lock of whoever is mapping / page-locking it

    def tensor(self, path: str, samples: int):
        """The arena as a page-locked int16 tensor of at least `samples` samples.  Page-locking -- 48 arenas of 16 MB: ~0.1 s,
        scripts/exp_arena_register.py -- belongs on a thread that is not the submitting one (BatchQueue.arena_ready, called by the thread
        that waits for the readers as soon as an answer names the arena); hence the lock.  (Page-locking all of them ahead, while the
        loci are set up, was slower: 48 x 16 MB took 0.3 s beside sixteen set-up threads and sixteen readers on sixteen CPUs.)"""
        import mmap
        torch = self.torch
        got = self.entries.get(path)
        if got is not None and got[1].numel() >= samples:
            return got[1]   # (without the lock: another thread may be page-locking ANOTHER arena)
        with self.lock:
            mine = self.busy.setdefault(path, threading.Lock())
        with mine:
            got = self.entries.get(path)
            if got is None or got[1].numel() < samples:
                if got is not None:
                    self._unregister(got)
                with open(path, 'r+b') as fh:
                    size = os.fstat(fh.fileno()).st_size
                    mm = mmap.mmap(fh.fileno(), size)   # (not pre-faulted: page-locking maps the pages, and faster)
                view = np.frombuffer(mm, dtype=np.int16)
                registered = False
                try:
                    with torch.cuda.device(self.dev):
                        registered = int(torch.cuda.cudart().cudaHostRegister(view.ctypes.data, size, 0)) == 0
                except Exception:  # noqa: BLE001 -- a pageable buffer uploads as well, only slower
                    pass
                got = (mm, torch.from_numpy(view), registered, view.ctypes.data)
                with self.lock:
                    self.entries[path] = got
            return got[1]

    def _unregister(self, got):
        if got[2]:
            try:
                self.torch.cuda.cudart().cudaHostUnregister(got[3])
            except Exception:  # noqa: BLE001
                pass

    def close(self):
        """Release every arena from page-locking (a millisecond each, and one after the other whoever asks: eight threads side by
        side took as long)."""
        with self.lock:
            for got in self.entries.values():
                self._unregister(got)
            self.entries.clear()


class BatchQueue:
    """Back-to-back batches on one handle without draining the GPU in between: submit() enqueues the upload, the signal
    loader and the caller of a batch and returns at once; collect() waits for that batch only and returns its records and
    called sequences (packed on the device: what comes down is sum(len) bytes, not a byte per sample).  While batch k runs,
    the host reads the files of batch k+1 -- upstream's per-locus loop (WarpSTR.py:33-76) has nothing to overlap, here the
    loci of a run share the GPU.  torch is the device allocator and the stream owner, nothing else.

    The handle must have been created on `stream` (a torch.cuda.Stream): inputs are uploaded there, and the library reads a
    call's inputs in the order of the handle's stream (wsx_caller_set_pipelined)."""

    def __init__(self, hip: 'HipCaller', stream, spike_removal: str = 'Brute'):
        import torch
        if spike_removal not in SPIKE_REMOVAL:
            raise ValueError(f'spike_removal must be one of {sorted(SPIKE_REMOVAL)}')
        self.torch, self.hip, self.stream, self.spike = torch, hip, stream, spike_removal
        self.dev = torch.device('cuda', hip.device)
        self.down = torch.cuda.Stream(device=self.dev)  # ordered after every call so far (wsx_caller_join): the records come down here
        self.pack = torch.cuda.Stream(device=self.dev)  # packs a finished batch's sequences: waits for THAT batch only
        self.up = torch.cuda.Stream(device=self.dev)    # uploads of the reader arenas' bytes (submit_vbz_parts), beside the kernels
        self._staging = {}   # dtype -> list of [pinned tensor, event of its last upload]
        self._turn = 0
        self._shared = None  # SharedStaging: buffers the reader processes fill (stage_shared)
        self.arenas = ArenaTable(self.dev)   # the readers' arenas, mapped and page-locked
        self.parts_s = {}    # where submit_raw_parts spent its time (seconds, summed over the batches)
        self.collect_parts = {}   # ... and collect: waiting for the batch, packing its sequences
        self._region_events = {}   # arena region -> event of the upload that last read it
        hip.set_pipelined(True)

    def _stage(self, dtype, count: int):
        """A pinned host buffer of `count` elements whose previous upload has left it (three take turns)."""
        torch = self.torch
        ring = self._staging.setdefault(dtype, [[None, None] for _ in range(3)])
        slot = ring[self._turn % len(ring)]
        self._turn += 1
        if slot[1] is not None:
            slot[1].synchronize()
        if slot[0] is None or slot[0].numel() < count:
            slot[0] = torch.empty(max(count + count // 4, 1 << 16), dtype=dtype, pin_memory=True)
        return slot

    def _launch(self, n, offsets, aut, signal, keep):
        torch = self.torch
        total = int(offsets[-1])
        with torch.cuda.stream(self.stream):
            records = torch.zeros((max(n, 1), _lib.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=self.dev)
            seq1 = torch.empty(max(total, 1), dtype=torch.uint8, device=self.dev)
            seq2 = torch.empty(max(total, 1), dtype=torch.uint8, device=self.dev)
        self.hip.call_device(signal.data_ptr(), offsets, aut, records.data_ptr(), seq1_ptr=seq1.data_ptr(), seq2_ptr=seq2.data_ptr())
        self.hip.join(self.down.cuda_stream)
        rec_host = torch.empty((max(n, 1), _lib.RESULT_DTYPE.itemsize), dtype=torch.uint8, pin_memory=True)
        with torch.cuda.stream(self.down):
            rec_host.copy_(records, non_blocking=True)
            done = torch.cuda.Event()
            done.record()
        return dict(n=n, offsets=offsets, records=records, seq1=seq1, seq2=seq2, rec_host=rec_host, done=done, keep=(signal, keep))

    def submit_raw(self, raws: Sequence[np.ndarray], lo: np.ndarray, hi: np.ndarray, aut: np.ndarray):
        """raws[r]: the whole raw read (int16 DAC values); [lo[r], hi[r]]: its STR segment (overview columns l_start_raw,
        r_end_raw); aut[r]: automaton of the handle.  Spike removal, whole-read normalisation and the slice run on the GPU."""
        torch = self.torch
        n = len(raws)
        # the reads' own buffers go to the page-locked staging ring: addresses by a C loop over the list (csrc/seam_helper.c), the
        # copies on a few threads without the GIL (csrc/host_loci.cpp) -- 50 000 reads: 35 ms as one np.concatenate
        ptrs, seam = None, _seam()
        if seam is not None and n:
            from . import _hostlib
            if _hostlib.lib() is not None:
                ptrs, lens = np.zeros(n, np.uintp), np.zeros(n, np.int64)
                if seam.wsx_seam_buffers_i16(raws, _lib.ptr(ptrs), _lib.ptr(lens)) != n:
                    ptrs = None
        if ptrs is None:
            lens = np.fromiter((len(r) for r in raws), dtype=np.int64, count=n)
        roff = np.zeros(n + 1, np.int64)
        np.cumsum(lens, out=roff[1:])
        total_raw = int(roff[-1])
        slot = self._stage(torch.int16, total_raw)
        if ptrs is not None:
            nbytes = lens * 2
            _hostlib.lib().wsh_gather(_lib.ptr(ptrs), _lib.ptr(nbytes), n, slot[0].data_ptr(), min(8, os.cpu_count() or 1))
        elif n:
            np.concatenate([np.asarray(r, np.int16) for r in raws], out=slot[0].numpy()[:total_raw])
        lo, hi = np.ascontiguousarray(lo, np.int64), np.ascontiguousarray(hi, np.int64)
        offsets = np.zeros(n + 1, np.int64)
        np.cumsum(slice_lengths(lo, hi, lens), out=offsets[1:])
        with torch.cuda.stream(self.stream):
            raw_dev = torch.empty(max(total_raw, 1), dtype=torch.int16, device=self.dev)
            raw_dev[:total_raw].copy_(slot[0][:total_raw], non_blocking=True)
            slot[1] = torch.cuda.Event()
            slot[1].record()
            signal = torch.empty(max(int(offsets[-1]), 1), dtype=torch.float64, device=self.dev)
        if n:
            self.hip.prepare_device(raw_dev.data_ptr(), roff, lo, hi, signal.data_ptr(), offsets, self.spike)
        return self._launch(n, offsets, np.ascontiguousarray(aut, np.int32), signal, raw_dev)

    def stage_shared(self, count: int) -> dict:
        """A staging slot of `count` int16 samples that reader processes can fill by path (SharedStaging), page-locked for the
        upload where the runtime allows it; submit_raw_shared() takes it."""
        if self._shared is None:
            torch = self.torch
            rt = torch.cuda.cudart()

            def register(address, nbytes):
                try:
                    with torch.cuda.device(self.dev):   # (may be called from the driver's reader thread)
                        return int(rt.cudaHostRegister(address, nbytes, 0)) == 0
                except Exception:  # noqa: BLE001 -- a pageable buffer uploads as well, only slower
                    return False

            def unregister(address):
                try:
                    rt.cudaHostUnregister(address)
                except Exception:  # noqa: BLE001
                    pass
            self._shared = SharedStaging(3, register, unregister)
        return self._shared.take(count)

    def stage_local(self, count: int) -> dict:
        """A staging slot of `count` int16 samples in this process's page-locked ring (the one submit_raw concatenates into), for
        a reader that decodes its reads straight to their places; submit_raw_shared() takes it."""
        pinned = self._stage(self.torch.int16, count)
        return dict(path=None, view=pinned[0].numpy(), tensor=pinned[0], pinned=pinned, event=None)

    def submit_raw_shared(self, slot: dict, roff: np.ndarray, lo: np.ndarray, hi: np.ndarray, aut: np.ndarray):
        """submit_raw() for reads that lie back to back in a staging slot already (read r = slot['view'][roff[r] : roff[r + 1]])."""
        torch = self.torch
        n = len(roff) - 1
        roff = np.ascontiguousarray(roff, np.int64)
        lens = np.diff(roff)
        total_raw = int(roff[-1])
        lo, hi = np.ascontiguousarray(lo, np.int64), np.ascontiguousarray(hi, np.int64)
        offsets = np.zeros(n + 1, np.int64)
        np.cumsum(slice_lengths(lo, hi, lens), out=offsets[1:])
        host = slot['tensor'] if slot.get('tensor') is not None else torch.from_numpy(slot['view'])
        with torch.cuda.stream(self.stream):
            raw_dev = torch.empty(max(total_raw, 1), dtype=torch.int16, device=self.dev)
            raw_dev[:total_raw].copy_(host[:total_raw], non_blocking=True)
            slot['event'] = torch.cuda.Event()
            slot['event'].record()
            if slot.get('pinned') is not None:
                slot['pinned'][1] = slot['event']   # (the ring waits for this upload before it hands the buffer out again)
            signal = torch.empty(max(int(offsets[-1]), 1), dtype=torch.float64, device=self.dev)
        if n:
            self.hip.prepare_device(raw_dev.data_ptr(), roff, lo, hi, signal.data_ptr(), offsets, self.spike)
        return self._launch(n, offsets, np.ascontiguousarray(aut, np.int32), signal, raw_dev)

    # ---- reader arenas (_readers.decode_arena): every reader process decodes its chunks into memory-backed files of its own ----
    ARENA_REGIONS = int(os.environ.get('WARPSTR_ARENA_REGIONS', '3') or 3)   # a region is written again only after the batch that used it was uploaded

    def region_wait(self, region: int):
        """Block until the batch that last used `region` of the readers' arenas has been uploaded."""
        ev = self._region_events.get(region)
        if ev is not None:
            ev.synchronize()

    def _arena_tensor(self, path: str, samples: int):
        """A reader's arena file as a page-locked tensor (ArenaTable.tensor)."""
        return self.arenas.tensor(path, samples)

    def arena_is_ready(self, path: str, samples: int) -> bool:
        """Is the arena mapped and page-locked at that size already (nothing to do for arena_ready)?"""
        got = self.arenas.entries.get(path)
        return got is not None and got[1].numel() >= samples

    def arena_ready(self, path: str, samples: int):
        """Map and page-lock a reader's arena ahead of the submit_raw_parts() that uploads from it (any thread)."""
        self._arena_tensor(path, samples)

    def submit_raw_parts(self, region: int, parts, lo: np.ndarray, hi: np.ndarray, aut: np.ndarray):
        """submit_raw() for a batch whose reads lie in the readers' arenas: parts = [(arena path, its size in samples, start of the
        chunk, [length of every read]), ...] in batch order; one upload per chunk, straight from the page-locked arena."""
        import time
        torch = self.torch
        clock, acc = time.perf_counter, self.parts_s
        t0 = clock()
        lens = np.array([n for _, _, _, ls in parts for n in ls], np.int64)
        n = len(lens)
        roff = np.zeros(n + 1, np.int64)
        np.cumsum(lens, out=roff[1:])
        total_raw = int(roff[-1])
        lo, hi = np.ascontiguousarray(lo, np.int64), np.ascontiguousarray(hi, np.int64)
        offsets = np.zeros(n + 1, np.int64)
        np.cumsum(slice_lengths(lo, hi, lens), out=offsets[1:])
        t1 = clock()
        with torch.cuda.stream(self.stream):
            raw_dev = torch.empty(max(total_raw, 1), dtype=torch.int16, device=self.dev)
            t2 = clock()
            at = 0
            for path, cap, base, ls in parts:
                m = int(sum(ls))
                if m:
                    raw_dev[at:at + m].copy_(self._arena_tensor(path, cap)[base:base + m], non_blocking=True)
                at += m
            ev = torch.cuda.Event()
            ev.record()
            self._region_events[region] = ev
            t3 = clock()
            signal = torch.empty(max(int(offsets[-1]), 1), dtype=torch.float64, device=self.dev)
        t4 = clock()
        if n:
            self.hip.prepare_device(raw_dev.data_ptr(), roff, lo, hi, signal.data_ptr(), offsets, self.spike)
        t5 = clock()
        ticket = self._launch(n, offsets, np.ascontiguousarray(aut, np.int32), signal, raw_dev)
        for key, dt in (('lengths', t1 - t0), ('allocate', t2 - t1 + t4 - t3), ('copies', t3 - t2), ('prepare', t5 - t4), ('launch', clock() - t5)):
            acc[key] = acc.get(key, 0.0) + dt
        return ticket

    DEVICE_ZSTD = True   # submit_vbz_parts also takes chunks whose zstd frame is still around them (wsx_zstd_decode undoes it)

    def submit_vbz_parts(self, region: int, parts, lo: np.ndarray, hi: np.ndarray, aut: np.ndarray):
        """submit_raw_parts() for readers that leave the decoding to the device (_readers.pack_arena): parts = [(arena path, its
        size in bytes, first byte of the chunk, bytes used, [samples of every read], the blocks' seven int64 as bytes), ...] in
        batch order.  One upload per chunk of what the readers wrote -- the zstd frames as they lie in the file (half the
        samples' bytes) or, where a reader undid zstd, StreamVByte blocks --, then wsx_zstd_decode (frames -> StreamVByte blocks),
        wsx_vbz_decode (-> the batch's int16 buffer), then the signal loader and the caller as ever."""
        import time
        torch = self.torch
        clock, acc = time.perf_counter, self.parts_s
        t0 = clock()
        lens = np.array([n for p in parts for n in p[4]], np.int64)
        n = len(lens)
        roff = np.zeros(n + 1, np.int64)
        np.cumsum(lens, out=roff[1:])
        total_raw = int(roff[-1])
        lo, hi = np.ascontiguousarray(lo, np.int64), np.ascontiguousarray(hi, np.int64)
        offsets = np.zeros(n + 1, np.int64)
        np.cumsum(slice_lengths(lo, hi, lens), out=offsets[1:])
        # the block table: where a block lies in the batch's byte buffer, where its samples go -- for all parts at once (a batch has
        # two chunks per reader: per-part NumPy calls were a tenth of a second of a 0.85 s run)
        n_parts = len(parts)
        used_p = np.fromiter((p[3] for p in parts), np.int64, n_parts)
        base_p = np.fromiter((p[2] for p in parts), np.int64, n_parts)
        reads_p = np.fromiter((len(p[4]) for p in parts), np.int64, n_parts)
        at_p = np.zeros(n_parts + 1, np.int64)
        np.cumsum((used_p + 15) & ~15, out=at_p[1:])
        first_p = np.cumsum(reads_p) - reads_p
        at_src = int(at_p[-1])
        spans = [(p[0], p[1], p[2], p[3], int(at_p[k])) for k, p in enumerate(parts)]
        t = np.frombuffer(b''.join(p[5] for p in parts), np.int64).reshape(-1, 7)
        blocks = np.zeros(len(t), _lib.VBZ_BLOCK_DTYPE)
        if len(t):
            part_of = np.repeat(np.arange(n_parts), np.fromiter((len(p[5]) // 56 for p in parts), np.int64, n_parts))
            read_of = first_p[part_of] + t[:, 0]
            blocks['src_offset'] = at_p[part_of] + (t[:, 2] - base_p[part_of])
            blocks['src_bytes'], blocks['n_samples'], blocks['kind'], blocks['n_values'] = t[:, 3], t[:, 4], t[:, 1], t[:, 5]
            # samples of the read's earlier blocks (a read is several blocks when its dataset has several chunks)
            before = np.cumsum(t[:, 4]) - t[:, 4]
            starts = np.flatnonzero(np.r_[True, read_of[1:] != read_of[:-1]])
            before -= np.repeat(before[starts], np.diff(np.r_[starts, len(t)]))
            blocks['dst_offset'] = roff[read_of] + before
        content = t[:, 6]
        # blocks that are still zstd frames (kinds 3 / 4): their content gets a place behind the uploaded bytes, in the same buffer
        framed = np.flatnonzero(blocks['kind'] >= 3)
        frames, svb_at = np.zeros(len(framed), _lib.ZSTD_FRAME_DTYPE), at_src
        if len(framed):
            room = (content[framed] + 15) & ~15
            place = svb_at + np.cumsum(room) - room
            frames['src_offset'], frames['src_bytes'] = blocks['src_offset'][framed], blocks['src_bytes'][framed]
            frames['dst_offset'], frames['dst_bytes'] = place - svb_at, content[framed]
            blocks['src_offset'][framed], blocks['src_bytes'][framed] = place, content[framed]
            blocks['kind'][framed] -= 2
            svb_bytes = int(room.sum())
        else:
            svb_bytes = 0
        t1 = clock()
        # The upload on a stream of its own, into a buffer that BELONGS to that stream (allocated under it): it runs beside the kernels
        # of the batch before.  (A buffer allocated in the handle's stream order may be memory an earlier batch's kernels still use,
        # so the copies had to wait for everything enqueued there -- the batch before included: 3 ms of copies between two
        # batches' kernels, profiles/r06_from_fast5_60k_trace.log.)  The decoders wait for the copies; the allocator hears that the
        # handle's stream uses the buffer, and hands it out again only when that stream has passed the point where it was dropped.
        with torch.cuda.stream(self.up):
            src_dev = torch.empty(max(at_src + svb_bytes, 16), dtype=torch.uint8, device=self.dev)
            src_dev.record_stream(self.stream)
            t2 = clock()
            for path, cap, base, used, at in spans:
                if used:
                    src_dev[at:at + used].copy_(self._arena_tensor(path, cap // 2).view(torch.uint8)[base:base + used], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        t3 = clock()
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev)
            self._region_events[region] = ev
            lit_dev = torch.empty(max(svb_bytes, 16), dtype=torch.uint8, device=self.dev) if svb_bytes else None
            raw_dev = torch.empty(max(total_raw, 1), dtype=torch.int16, device=self.dev)
            status = torch.empty(max(len(blocks), 1) + max(len(frames), 1), dtype=torch.int32, device=self.dev)
            signal = torch.empty(max(int(offsets[-1]), 1), dtype=torch.float64, device=self.dev)
        t4 = clock()
        zst = status[max(len(blocks), 1):]
        if len(frames):
            self.hip.zstd_decode_device(src_dev.data_ptr(), at_src, frames, src_dev.data_ptr() + svb_at, svb_bytes, lit_dev.data_ptr(), zst.data_ptr())
        if len(blocks):
            self.hip.vbz_decode_device(src_dev.data_ptr(), at_src + svb_bytes, blocks, raw_dev.data_ptr(), max(total_raw, 1), status.data_ptr())
        if n:
            self.hip.prepare_device(raw_dev.data_ptr(), roff, lo, hi, signal.data_ptr(), offsets, self.spike)
        t5 = clock()
        with torch.cuda.stream(self.stream):   # (status was allocated on this stream: its memory is reused in its order)
            bad = None
            if len(blocks):
                flag = status[:len(blocks)].ne(0).any()
                if len(frames):
                    flag = flag | zst[:len(frames)].ne(0).any()
                # (on its way to the host with the batch: `done` of the ticket covers the copy, collect reads a byte of host memory)
                bad = torch.empty(1, dtype=torch.bool, pin_memory=True)
                bad.copy_(flag.reshape(1), non_blocking=True)
        ticket = self._launch(n, offsets, np.ascontiguousarray(aut, np.int32), signal, (raw_dev, lit_dev))
        ticket['vbz_bad'] = bad
        self.vbz_bytes = getattr(self, 'vbz_bytes', 0) + at_src
        self.zstd_frames = getattr(self, 'zstd_frames', 0) + len(frames)
        for key, dt in (('lengths', t1 - t0), ('allocate', t2 - t1 + t4 - t3), ('copies', t3 - t2), ('prepare', t5 - t4), ('launch', clock() - t5)):
            acc[key] = acc.get(key, 0.0) + dt
        return ticket

    def close(self):
        if self._shared is not None:
            self._shared.close()
            self._shared = None
        for ev in self._region_events.values():
            ev.synchronize()
        self._region_events.clear()
        self.arenas.close()

    def submit_signals(self, signals: Sequence[np.ndarray], aut: np.ndarray):
        """Already normalised float64 segments (a `signal_loader`, or ReadSignal workloads)."""
        torch = self.torch
        n = len(signals)
        lens = np.fromiter((len(x) for x in signals), dtype=np.int64, count=n)
        offsets = np.zeros(n + 1, np.int64)
        np.cumsum(lens, out=offsets[1:])
        total = int(offsets[-1])
        slot = self._stage(torch.float64, total)
        if n:
            np.concatenate([np.asarray(x, np.float64) for x in signals], out=slot[0].numpy()[:total])
        with torch.cuda.stream(self.stream):
            signal = torch.empty(max(total, 1), dtype=torch.float64, device=self.dev)
            signal[:total].copy_(slot[0][:total], non_blocking=True)
            slot[1] = torch.cuda.Event()
            slot[1].record()
        return self._launch(n, offsets, np.ascontiguousarray(aut, np.int32), signal, None)

    def collect(self, ticket):
        """-> (records, seq1, pos1, seq2, pos2): read r's seq is seq1[pos1[r] : pos1[r + 1]] (empty for a failed read)."""
        torch = self.torch
        import time
        n = ticket['n']
        clock = time.perf_counter
        acc = self.collect_parts
        t0 = clock()
        ticket['done'].synchronize()
        t1 = clock()
        acc['wait'] = acc.get('wait', 0.0) + t1 - t0
        if ticket.get('vbz_bad') is not None and bool(ticket['vbz_bad'].item()):
            # (the readers check every block before it is uploaded: this is a block that changed on its way, not a bad file)
            raise RuntimeError('the device decoders flagged a chunk: a zstd frame that is corrupt (wsx_zstd_decode) or a StreamVByte block whose '
                               'keys ask for more bytes than it has (wsx_vbz_decode)')
        rec = ticket['rec_host'].numpy().view(_lib.RESULT_DTYPE).reshape(-1)[:n].copy()
        ok = rec['status'] == 0
        out = [rec]
        starts = torch.from_numpy(ticket['offsets'][:-1].copy())
        packed = []
        self.pack.wait_event(ticket['done'])  # (not the `down` stream itself: later batches have joined it since)
        with torch.cuda.stream(self.pack):
            starts_d = starts.to(self.dev, non_blocking=True)
            for field, key in (('len1', 'seq1'), ('len2', 'seq2')):
                ln = np.where(ok, rec[field], 0).astype(np.int64)
                pos = np.zeros(n + 1, np.int64)
                np.cumsum(ln, out=pos[1:])
                total = int(pos[-1])
                host = torch.empty(max(total, 1), dtype=torch.uint8, pin_memory=True)
                if total:
                    ln_d = torch.from_numpy(ln).to(self.dev, non_blocking=True)
                    pos_d = torch.from_numpy(pos[:-1].copy()).to(self.dev, non_blocking=True)
                    idx = torch.repeat_interleave(starts_d - pos_d, ln_d, output_size=total) + torch.arange(total, device=self.dev)
                    host[:total].copy_(ticket[key][idx], non_blocking=True)
                packed.append((host, pos, total))
            self.pack.synchronize()
        for host, pos, total in packed:
            out += [host.numpy()[:total].copy(), pos]
        acc['pack'] = acc.get('pack', 0.0) + clock() - t1
        return tuple(out)


def sequence_from_trace(table: AutomatonTable, flank_length: int, trace: np.ndarray, reverse: bool) -> str:
    """WarpSTR._get_sequence (src/caller/caller.py:178-187): last base of every visited state, flanks
    stripped with the reference's slice arithmetic, reverse-complemented for reverse-strand reads."""
    trace = np.asarray(trace)
    starts = np.insert(np.diff(trace) != 0, 0, True)
    trans = trace[starts]
    seq = table.last_base[trans].tobytes().decode('ascii')
    offset = int(table.seq_idx[trans[0]])
    seq = seq[flank_length - offset:-flank_length]
    return seq.translate(_COMPLEMENT)[::-1] if reverse else seq


def similarity_report(sequence: str, pore_model, min_state_similarity: float):
    """(text of summaries/state_similarity.csv, upstream's warning lines, (template_problems, reverse_problems)) of a locus
    pattern: CallerWrapper.check_high_similarity (src/caller/wrapper.py:122-160) without the side effects."""
    from .automata import reverse_pattern
    diffs_t = pore_model.get_diffs_for_all(sequence)
    diffs_r = pore_model.get_diffs_for_all(reverse_pattern(sequence))
    lines = ['pattern,strand,mean_diff,median_diff\n']
    lines += [f'{k},template,{mean:.3f},{med:.3f}\n' for k, (mean, med) in diffs_t.items()]
    lines += [f'{k},reverse,{mean:.3f},{med:.3f}\n' for k, (mean, med) in diffs_r.items()]
    lim = min_state_similarity
    problems = [[dict(pattern=k, mean_diff=mean, median_diff=med) for k, (mean, med) in d.items() if lim > mean or lim > med]
                for d in (diffs_t, diffs_r)]
    warnings = ['Warning: Template has repeat unit {} with high state similarity'.format(p['pattern']) for p in problems[0]]
    warnings += ['Warning: high similarity of state values in reverse pattern {}'.format(p['pattern']) for p in problems[1]]
    return ''.join(lines), warnings, (problems[0], problems[1])


class CallerWrapper:
    """Step-3 driver for one locus (src/caller/wrapper.py:57-120).

    Two call shapes:
      CallerWrapper(locus, threads)  -- the reference's own (wrapper.py:63-70): `locus` is any object with `.path`,
          `.sequence` and `.flank_length` (src/schemas/locus.py); the flanks are read from
          <locus.path>/expected_signals/sequences.csv and summaries/state_similarity.csv is written, as upstream does.
      CallerWrapper(sequence, flanks, flank_length, threads=1, ...)  -- without a locus directory;
          ``flanks`` = (left_template, right_template, left_reverse, right_reverse).
    ``threads`` is accepted for interface compatibility; reads are fanned out over GPU wavefronts.
    """

    def __init__(self, locus_or_sequence, flanks_or_threads=None, flank_length: Optional[int] = None, threads: int = 1,
                 caller_config: Optional[CallerConfig] = None, rescaler_config: Optional[RescalerConfig] = None,
                 device: int = 0, pore_model=None, on_error: str = 'raise', flanks: Optional[Sequence[str]] = None,
                 write_summaries: bool = True):
        """write_summaries=False: the locus directory is left alone (ranks other than 0 of a multi-GPU job)."""
        from .pore_model import default_pore_model
        self._write_summaries = bool(write_summaries)
        self.caller_config = caller_config or CallerConfig()
        self.pore_model = pore_model or default_pore_model()
        if isinstance(locus_or_sequence, str):
            self.locus = None
            self.sequence = locus_or_sequence.upper()
            flanks = flanks if flanks is not None else flanks_or_threads
            self.flank_length = int(flank_length)
            self.threads = threads
        else:
            from . import overview as ov
            self.locus = locus_or_sequence
            self.sequence = self.locus.sequence.upper()
            self.flank_length = int(self.locus.flank_length)
            self.threads = int(flanks_or_threads) if flanks_or_threads is not None else threads
            flanks = ov.load_flanks(self.locus.path)
            self.check_high_similarity(self.sequence)
        self.on_error = on_error
        lt, rt, lr, rr = flanks
        self.temp_sta, self.rev_sta = locus_automata(lt, rt, lr, rr, self.sequence, pore_model)
        self.hip = HipCaller([self.temp_sta, self.rev_sta], [self.flank_length, self.flank_length], self.caller_config,
                             rescaler_config, device=device)

    def check_high_similarity(self, sequence: str, out_dir: Optional[str] = None):
        """Similarity of consecutive state levels per repeat unit (src/caller/wrapper.py:122-160): writes
        <locus.path>/summaries/state_similarity.csv (same columns and %.3f formatting), prints upstream's warnings for
        units whose mean or median level difference is below caller_config.min_state_similarity, and returns
        (template_problems, reverse_problems)."""
        text, warnings, problems = similarity_report(sequence, self.pore_model, self.caller_config.min_state_similarity)
        if out_dir is None and self.locus is not None and self._write_summaries:
            out_dir = os.path.join(self.locus.path, 'summaries')
        if out_dir is not None:
            os.makedirs(out_dir, exist_ok=True)
            with open(os.path.join(out_dir, 'state_similarity.csv'), 'w') as f:
                f.write(text)
        for line in warnings:
            print(line)
        return problems

    def run(self, workload: List[ReadSignal]) -> 'CallerResults':
        """Run the caller for each piece of signal in the workload; results align with the workload."""
        if not workload:
            return []
        # (no packing on this side: the library gathers the reads' own arrays into its upload buffers)
        res, offsets, extra = self.hip.call_workload(workload, want_seqs=True)
        return CallerResults(_WorkloadNames(workload), res, extra['seq1_pos'][:-1], extra['seq1'], extra['seq2'], self.on_error,
                             offsets2=extra['seq2_pos'][:-1]).check()

    def run_raw(self, names: Sequence[str], reverses: Sequence[bool], raws: Sequence[np.ndarray],
                positions: Sequence[Sequence[int]], spike_removal: str = 'Brute') -> 'CallerResults':
        """Same as run(), from raw DAC reads: spike removal, whole-read normalisation and the slice
        [l_start_raw, r_end_raw] (Fast5.get_data_processed, src/schemas/fast5.py:45-57) happen on the GPU."""
        if not names:
            return []
        if spike_removal not in SPIKE_REMOVAL:
            raise ValueError(f'spike_removal must be one of {sorted(SPIKE_REMOVAL)}')
        # The reads cross PCIe once, as int16 (2 bytes per sample of the WHOLE read go up, the records and the called
        # sequences come down); the normalised float64 segments are produced and consumed in HBM.  torch is the device
        # allocator here (pinned staging buffer, HBM buffers), nothing else.
        try:
            import torch
            have_torch = torch.cuda.is_available()
        except ImportError:
            have_torch = False
        if not have_torch:  # a process without (a GPU build of) torch: the same through the library's host-buffer entry points
            signal, offsets, _ = self.hip.prepare_signals(raws, positions, spike_removal)
            return self._run_packed(list(names), reverses, signal, offsets)
        dev = torch.device('cuda', self.hip.device)
        n = len(names)
        lens = np.fromiter((len(r) for r in raws), dtype=np.int64, count=n)
        roff = np.zeros(n + 1, np.int64)
        np.cumsum(lens, out=roff[1:])
        staged = torch.empty(int(roff[-1]), dtype=torch.int16, pin_memory=True)
        raw_host = staged.numpy()
        for r, o in zip(raws, roff[:-1]):
            raw_host[o:o + len(r)] = r
        lo = np.array([p[0] for p in positions], np.int64)
        hi = np.array([p[1] for p in positions], np.int64)
        seglen = np.array([len(range(*slice(int(a), int(b) + 1).indices(int(L)))) for a, b, L in zip(lo, hi, lens)], np.int64)
        offsets = np.zeros(n + 1, np.int64)
        np.cumsum(seglen, out=offsets[1:])
        total = int(offsets[-1])
        aut = np.fromiter((1 if r else 0 for r in reverses), dtype=np.int32, count=n)
        with torch.cuda.device(dev):
            raw_dev = staged.to(dev, non_blocking=True)
            signal = torch.empty(max(total, 1), dtype=torch.float64, device=dev)
            records = torch.zeros((n, _lib.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
            seq1 = torch.zeros(max(total, 1), dtype=torch.uint8, device=dev)
            seq2 = torch.zeros(max(total, 1), dtype=torch.uint8, device=dev)
            torch.cuda.current_stream(dev).synchronize()  # the handle's stream is not torch's: inputs are complete before it starts
            self.hip.prepare_device(raw_dev.data_ptr(), roff, lo, hi, signal.data_ptr(), offsets, spike_removal)
            self.hip.call_device(signal.data_ptr(), offsets, aut, records.data_ptr(), seq1_ptr=seq1.data_ptr(), seq2_ptr=seq2.data_ptr())
            self.hip.synchronize()
            res = records.cpu().numpy().view(_lib.RESULT_DTYPE).reshape(n)
            out = CallerResults(list(names), res, offsets[:-1], seq1[:total].cpu().numpy(), seq2[:total].cpu().numpy(), self.on_error)
        return out.check()

    def _run_packed(self, names, reverses, signal, offsets) -> 'CallerResults':
        aut = np.fromiter((1 if r else 0 for r in reverses), dtype=np.int32, count=len(names))
        res, extra = self.hip.call(signal, offsets, aut, want_seqs=True)
        out = CallerResults(names, res, offsets[:-1], extra['seq1'], extra['seq2'], self.on_error)
        return out.check()
