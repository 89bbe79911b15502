"""Builds the HIP library (warpstr_amd/libwarpstr_hip.so) in-tree for gfx950.

    python -m warpstr_amd.build [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off is REQUIRED: the caller reproduces the
reference's fp64 arithmetic operation by operation (HIP's default would fuse a*b+c into FMAs).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
SOURCES = ['wsx_api.hip', 'dtw_kernels.hip', 'mid_kernels.hip', 'wsx_prep.hip', 'wsx_vbz.hip', 'wsx_zstd.hip', 'flank_kernels.hip']
HEADERS = ['wsx_device.h', 'wsx_place.h', os.path.join('..', '..', 'include', 'warpstr_hip.h')]
LIB = os.path.join(HERE, 'libwarpstr_hip.so')
SEAM_SRC = os.path.join(CSRC, 'seam_helper.c')  # CPython-API loops of the Python seam (no compute); optional at run time
SEAM_LIB = os.path.join(HERE, '_seam_helper.so')
HOST_SRC = os.path.join(CSRC, 'host_loci.cpp')   # the per-locus host work (overview.csv, automata, output files): plain C++, no HIP
READER_SRC = os.path.join(CSRC, 'host_reader.cpp')   # ... and the fast5 reader loop of the reader processes (libhdf5 / libzstd by dlopen)
HOST_LIB = os.path.join(HERE, '_host_loci.so')
FLAGS = ['-O3', '--offload-arch=gfx950', '-ffp-contract=off', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function',
         '-fgpu-rdc' if False else '-fno-gpu-rdc']


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace('.hip', '.o'))
        objs.append(o)
        if force or _stale(o, [s] + hdrs + [os.path.abspath(__file__)]):
            cmd = [hipcc] + FLAGS + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd))
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    if force or _stale(SEAM_LIB, [SEAM_SRC]):
        import sysconfig
        cmd = [os.environ.get('CC', 'gcc'), '-O2', '-shared', '-fPIC', '-I', sysconfig.get_paths()['include'], SEAM_SRC, '-o', SEAM_LIB]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    if force or _stale(HOST_LIB, [HOST_SRC, READER_SRC]):
        cmd = [os.environ.get('CXX', 'g++'), '-O2', '-std=c++17', '-ffp-contract=off', '-shared', '-fPIC', '-fvisibility=hidden', '-Wall', '-pthread',
               HOST_SRC, READER_SRC, '-o', HOST_LIB, '-ldl']
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
