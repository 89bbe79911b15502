"""Properties of the compiled kernels that the source does not show and a change can lose without a test failing (hipcc
cross-compiles here: no GPU needed).
* A load through a pointer made from an integer is a FLAT load; a flat load counts as an LDS operation too, so loads that are meant to
  be in flight while a kernel works through LDS are waited for at its first wait for LDS (round 6: 3.03 against 2.79 ms in
  zstd_literals_kernel; the window loads of vbz_decode_kernel had been flat since round 5).
* A per-thread array that is indexed at run time is scratch memory, which the runtime backs with a large allocation per queue and
  takes back under memory pressure (round 6: zstd_index_kernel, 784 bytes a lane)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason='no hipcc on this machine')


def _assembly(name, tmp_path):
    out = str(tmp_path / (name + '.s'))
    subprocess.run([HIPCC, '-O3', '--offload-arch=gfx950', '-ffp-contract=off', '-std=c++17', '-fno-gpu-rdc', '-S', '--cuda-device-only',
                    os.path.join(ROOT, 'warpstr_amd', 'csrc', name + '.hip'), '-o', out], check=True, capture_output=True)
    text = open(out).read()
    kernels = {}
    for piece in re.split(r'\n(?=_Z[^\n]*:\s*;? *@)', text):
        sym = piece.split(':', 1)[0]
        if sym.startswith('_Z'):
            kernels[sym] = piece
    scratch = {m.group(1): int(m.group(2)) for m in re.finditer(r'\.amdhsa_kernel (\S+)\n\s*\.amdhsa_group_segment_fixed_size \d+\n\s*\.amdhsa_private_segment_fixed_size (\d+)', text)}
    return kernels, scratch


def _one(kernels, part):
    hits = [k for k in kernels if part in k]
    assert len(hits) == 1, (part, hits)
    return kernels[hits[0]]


def test_the_decoders_loads_in_flight_are_global_and_nothing_lives_in_scratch(tmp_path):
    kernels, scratch = _assembly('wsx_zstd', tmp_path)
    body = _one(kernels, 'zstd_literals_kernel')
    assert 'flat_load' not in body and body.count('global_load_dwordx4') >= 8
    assert 'flat_load' not in _one(kernels, 'zstd_index_kernel') and 'flat_load' not in _one(kernels, 'zstd_order_kernel')
    assert len(scratch) == 4 and all(v == 0 for v in scratch.values()), scratch
    kernels, scratch = _assembly('wsx_vbz', tmp_path)
    body = _one(kernels, 'vbz_decode_kernel')
    assert 'flat_load' not in body and 'global_load_dwordx4' in body
    assert all(v == 0 for v in scratch.values()), scratch
