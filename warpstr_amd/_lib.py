"""ctypes binding of the C ABI in include/warpstr_hip.h (warpstr_amd/libwarpstr_hip.so).

There is deliberately no CPU fallback: if the HIP library is missing or no MI355X is visible the
caller raises.  (Building: ``python -m warpstr_amd.build``.)
"""
import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('WARPSTR_HIP_LIB') or os.path.join(_HERE, 'libwarpstr_hip.so')

WSX_MEM_HOST, WSX_MEM_DEVICE = 0, 1
READ_STATUS = {0: 'ok', 1: 'shape', 2: 'backtrack', 3: 'fit_points', 4: 'fit_order', 5: 'fit_smooth', 6: 'no_repeat',
               7: 'segment_range'}
EXPORTS = ['wsx_abi_version', 'wsx_device_count', 'wsx_last_error', 'wsx_caller_create', 'wsx_caller_destroy', 'wsx_caller_add_automata',
           'wsx_caller_set_workspace_limit', 'wsx_caller_get_workspace_limit', 'wsx_caller_set_tuning', 'wsx_caller_create_times', 'wsx_caller_set_streams', 'wsx_call_batch', 'wsx_call_batch_reads', 'wsx_warp_batch', 'wsx_prepare_signals',
           'wsx_caller_synchronize', 'wsx_caller_set_pipelined', 'wsx_caller_join', 'wsx_caller_timing_window',
           'wsx_caller_last_timing', 'wsx_caller_workspace', 'wsx_caller_fill_intervals', 'wsx_caller_kernel_name', 'wsx_locate_flanks', 'wsx_moves_to_raw', 'wsx_vbz_decode', 'wsx_zstd_decode']


# wsx_caller_set_tuning knobs (include/warpstr_hip.h: WSX_TUNE_*)
TUNING = {'stream_traceback_min': 1, 'borders_wave_below': 2, 'segment_two_kernels': 3, 'fill_blocks_per_cu': 4, 'chunks': 5,
          'small_pipe_samples': 6, 'calls_in_flight': 7, 'small_calls_in_flight': 8}


class WsxAutomaton(C.Structure):
    _fields_ = [('n_states', C.c_int32), ('endstate', C.c_int32), ('flank_length', C.c_int32), ('reverse', C.c_int32),
                ('value', C.c_void_p), ('seq_idx', C.c_void_p), ('pred_ptr', C.c_void_p), ('pred_idx', C.c_void_p),
                ('repeat_mask', C.c_void_p), ('last_base', C.c_void_p)]


class WsxParams(C.Structure):
    _fields_ = [('min_values_per_state', C.c_int32), ('states_in_segment', C.c_int32), ('threshold', C.c_double),
                ('max_std', C.c_double), ('method_median', C.c_int32), ('reps_as_one', C.c_int32)]


class WsxTraces(C.Structure):
    _fields_ = [('trace1', C.c_void_p), ('trace2', C.c_void_p), ('rescaled', C.c_void_p), ('badmask', C.c_void_p),
                ('seq1', C.c_void_p), ('seq2', C.c_void_p)]


# numpy view of wsx_result
RESULT_DTYPE = np.dtype([('status', np.int32), ('len1', np.int32), ('len2', np.int32), ('n_trans1', np.int32),
                         ('n_trans2', np.int32), ('reserved', np.int32), ('cost1', np.float64), ('cost2', np.float64),
                         ('dtw_end_cost1', np.float64), ('dtw_end_cost2', np.float64)], align=True)
assert RESULT_DTYPE.itemsize == 56


# numpy view of wsx_vbz_block (wsx_vbz_decode); kinds: WSX_VBZ_*
VBZ_BLOCK_DTYPE = np.dtype([('src_offset', np.int64), ('src_bytes', np.int64), ('dst_offset', np.int64), ('n_samples', np.int32),
                            ('kind', np.int32), ('n_values', np.int32), ('reserved', np.int32)])
assert VBZ_BLOCK_DTYPE.itemsize == 40
VBZ_PLAIN, VBZ_SVB_ZIGZAG, VBZ_SVB = 0, 1, 2
# numpy view of wsx_zstd_frame (wsx_zstd_decode)
ZSTD_FRAME_DTYPE = np.dtype([('src_offset', np.int64), ('src_bytes', np.int64), ('dst_offset', np.int64), ('dst_bytes', np.int64)])
assert ZSTD_FRAME_DTYPE.itemsize == 32


class WsxAlignScores(C.Structure):
    _fields_ = [('match', C.c_int32), ('mismatch', C.c_int32), ('gap_open', C.c_int32), ('gap_extend', C.c_int32)]


# numpy view of wsx_flank_hit
FLANK_HIT_DTYPE = np.dtype([(n, np.int32) for n in ('status', 'score', 'start', 'end', 'matches', 'span', 'row0', 'col0', 'row1',
                                                    'col1', 'gaps_text', 'gaps_pattern', 'raw_score', 'n_ops', 'n_best_cells',
                                                    'tie_steps')])
assert FLANK_HIT_DTYPE.itemsize == 64


class HipLibraryMissing(RuntimeError):
    pass


_lib = None


def _raise_hw_queue_limit():
    """The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  A wsx_caller uses up to
    ten streams and the application has its own; streams that share a queue serialise behind each other's event waits,
    which undoes the chunk overlap of pipelined calls.  Called from load(), i.e. when a caller is first
    created -- importing the package changes nothing.  Only effective before the HIP runtime initialises (nothing is
    touched once torch has initialised it); an explicit setting of the user's is kept; WARPSTR_KEEP_HW_QUEUES=1 opts out."""
    import sys
    if 'GPU_MAX_HW_QUEUES' in os.environ or os.environ.get('WARPSTR_KEEP_HW_QUEUES'):
        return
    torch = sys.modules.get('torch')
    if torch is not None and getattr(torch, 'cuda', None) is not None and torch.cuda.is_initialized():
        return
    os.environ['GPU_MAX_HW_QUEUES'] = '8'


def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own HIP/HSA runtime.  A process that initialises the system runtime first (through
    this library) and imports torch afterwards ends up with two runtimes and torch sees no GPU; with torch imported first
    this library simply binds to torch's copy (same soname) and both work.  Make the second order the only one: if torch
    is installed but not imported yet, load its runtime libraries before ours.  WARPSTR_HIP_SYSTEM_RUNTIME=1 disables it."""
    import importlib.util
    import sys
    if 'torch' in sys.modules or os.environ.get('WARPSTR_HIP_SYSTEM_RUNTIME'):
        return
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), 'lib')
    for name in ('libhsa-runtime64.so', 'libamdhip64.so'):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def load():
    """Load the HIP library; raises HipLibraryMissing if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    _raise_hw_queue_limit()
    _share_torch_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(f'{LIB_PATH} not found: build it with `python -m warpstr_amd.build` '
                                '(the caller has no CPU path)')
    lib = C.CDLL(LIB_PATH)
    lib.wsx_last_error.restype = C.c_char_p
    lib.wsx_caller_kernel_name.restype = C.c_char_p
    lib.wsx_caller_kernel_name.argtypes = [C.c_void_p, C.c_int32]
    lib.wsx_caller_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.wsx_caller_destroy.argtypes = [C.c_void_p]
    lib.wsx_zstd_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    lib.wsx_caller_add_automata.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
    lib.wsx_caller_destroy.restype = None
    lib.wsx_caller_set_workspace_limit.argtypes = [C.c_void_p, C.c_uint64]
    lib.wsx_caller_get_workspace_limit.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    lib.wsx_caller_set_tuning.argtypes = [C.c_void_p, C.c_int32, C.c_int64]
    lib.wsx_caller_create_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    lib.wsx_caller_set_streams.argtypes = [C.c_void_p, C.c_int32]
    lib.wsx_caller_synchronize.argtypes = [C.c_void_p]
    lib.wsx_caller_set_pipelined.argtypes = [C.c_void_p, C.c_int32]
    lib.wsx_caller_join.argtypes = [C.c_void_p, C.c_void_p]
    lib.wsx_caller_timing_window.argtypes = [C.c_void_p, C.c_int32]
    lib.wsx_caller_workspace.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
    lib.wsx_call_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                   C.c_void_p]
    lib.wsx_call_batch_reads.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    lib.wsx_warp_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.wsx_prepare_signals.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                        C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.wsx_caller_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32),
                                           C.POINTER(C.c_double)]
    lib.wsx_caller_fill_intervals.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
    lib.wsx_locate_flanks.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    lib.wsx_moves_to_raw.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    lib.wsx_vbz_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
    _lib = lib
    return lib


def last_error() -> str:
    return load().wsx_last_error().decode()


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f'{what} failed (code {rc}): {last_error()}')


def ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)
