"""ctypes front of oracle/zstd_oracle.c (a CPU restatement of a Zstandard frame decoder after RFC 8878) -- TEST INFRASTRUCTURE ONLY.
Pinned against libzstd itself by tests/test_zstd_oracle.py."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
ERRORS = {-1: 'truncated', -2: 'not a zstd frame', -3: 'unsupported', -4: 'corrupt', -5: 'no room'}


def lib():
    global _LIB
    if _LIB is None:
        subprocess.check_call(['make', '-C', HERE, '-s', 'libzstd_oracle.so'])
        _LIB = C.CDLL(os.path.join(HERE, 'libzstd_oracle.so'))
        _LIB.wso_zstd_decode.restype = C.c_int64
        _LIB.wso_zstd_decode.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int32)]
        _LIB.wso_zstd_content_size.restype = C.c_int64
        _LIB.wso_zstd_content_size.argtypes = [C.c_void_p, C.c_int64]
    return _LIB


def content_size(frame: bytes) -> int:
    """The content size the frame declares (-1: none declared)."""
    buf = np.frombuffer(frame, np.uint8)
    v = lib().wso_zstd_content_size(buf.ctypes.data, len(buf))
    if v < -1:
        raise ValueError(f'zstd oracle: {ERRORS.get(int(v) + 10, v)}')
    return int(v)


def decode(frame: bytes, cap: int = None, want_blocks: bool = False):
    """The content of one frame (bytes); ValueError with the oracle's reason on a frame it does not accept."""
    buf = np.frombuffer(bytes(frame), np.uint8)
    if cap is None:
        cap = content_size(frame)
        if cap < 0:
            cap = 64 * len(buf) + (1 << 20)
    out = np.empty(max(cap, 1), np.uint8)
    nb = C.c_int32()
    n = lib().wso_zstd_decode(buf.ctypes.data, len(buf), out.ctypes.data, cap, C.byref(nb))
    if n < 0:
        raise ValueError(f'zstd oracle: {ERRORS.get(int(n), n)}')
    res = out[:n].tobytes()
    return (res, int(nb.value)) if want_blocks else res
