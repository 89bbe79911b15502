"""The C-ABI library loads on a CPU-only machine and exports every symbol include/warpstr_hip.h declares.
No compute is attempted here (there is no CPU path): without a GPU, creation must fail loudly."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from warpstr_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, 'include', 'warpstr_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(wsx_[a-z_]+)\s*\(', text)))


@pytest.fixture(scope='module')
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        from warpstr_amd import build
        build.build(verbose=False)
    return _lib.load()


def test_exports_match_header(lib):
    declared = _declared_functions()
    assert declared and sorted(_lib.EXPORTS) == declared
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert lib.wsx_abi_version() == 14


def test_struct_layouts():
    assert C.sizeof(_lib.WsxAutomaton) == 64
    assert C.sizeof(_lib.WsxParams) == 32
    assert C.sizeof(_lib.WsxTraces) == 48
    assert _lib.RESULT_DTYPE.itemsize == 56
    assert [_lib.RESULT_DTYPE.fields[k][1] for k in ('status', 'len1', 'len2', 'cost1', 'dtw_end_cost2')] == \
        [0, 4, 8, 24, 48]
    assert C.sizeof(_lib.WsxAlignScores) == 16
    assert sorted(_lib.TUNING.values()) == list(range(1, 9))
    assert _lib.FLANK_HIT_DTYPE.itemsize == 64 and _lib.FLANK_HIT_DTYPE.fields['n_ops'][1] == 52 and _lib.FLANK_HIT_DTYPE.fields['tie_steps'][1] == 60


def test_no_gpu_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    assert lib.wsx_device_count() == 0
    from warpstr_amd import synth
    from warpstr_amd.caller import HipCaller
    locus = synth.make_locus('(AGC)', 16, 1)
    with pytest.raises(RuntimeError, match='no HIP device'):
        HipCaller([locus.template, locus.reverse], [16, 16])
    # and at the ABI level
    h = C.c_void_p()
    t = locus.template
    bufs = [np.ascontiguousarray(t.value), t.seq_idx, t.pred_ptr, t.pred_idx, t.repeat_mask, t.last_base]
    a = (_lib.WsxAutomaton * 1)(_lib.WsxAutomaton(t.n_states, t.endstate, 16, 0, *[_lib.ptr(b) for b in bufs]))
    p = _lib.WsxParams(4, 6, 0.5, 0.5, 0, 0)
    rc = lib.wsx_caller_create(C.byref(h), 0, C.byref(a), 1, C.byref(p), None)
    assert rc == -2 and b'no HIP device' in lib.wsx_last_error()
    # the flank entry points likewise
    from warpstr_amd import extractor
    with pytest.raises(RuntimeError, match='no HIP device'):
        extractor.locate(['ACGTACGT'], ['ACGT'])
    with pytest.raises(RuntimeError, match='no HIP device'):
        extractor.extract_from_moves_batch([np.ones(8, np.uint8)], [extractor.Position(1, 2)], [0], [5])


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under warpstr_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'warpstr_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                text = open(os.path.join(dirpath, f), errors='ignore').read()
                assert 'libwarpstr_oracle' not in text and 'libflank_oracle' not in text and 'from oracle' not in text \
                    and 'import oracle' not in text, f
