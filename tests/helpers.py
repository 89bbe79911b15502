"""Shared helpers for the parity tests (oracle side)."""
import os

import numpy as np

from oracle import oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
DEFAULT_CASES = ['agc_fl16', 'agc_fl29', 'aaat_fl110', 'hd_fl20', 'dm2_fl40', 'ngc_fl20', 'agc_fl16_ragged', 'real_aaat']


def load_case(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def golden_automaton(z, tag):
    """Oracle automaton from the tables stored in a fixture (tag 't' template / 'r' reverse)."""
    return oracle.Automaton(z[f'{tag}_value'], z[f'{tag}_seq_idx'], z[f'{tag}_pred_ptr'], z[f'{tag}_pred_idx'],
                            z[f'{tag}_mask'], int(z[f'{tag}_endstate']), int(z['flank_length']))


def assert_close_rel(a, b, rel=1e-5):
    a, b = float(a), float(b)
    if np.isnan(a) and np.isnan(b):
        return
    if np.isinf(a) or np.isinf(b):
        assert a == b
        return
    assert abs(a - b) <= rel * max(abs(a), abs(b), 1e-300), (a, b)


def write_perturbed_pore_model(path: str) -> str:
    """A pore-model table in upstream's format (example/deps/template_median68pA.model: tab-separated, columns kmer and
    level_mean among others) whose levels differ from the built-in r9.4 table -- a deterministic function of it -- for the tests
    of the `pore_model_path` configuration key (tests/golden/cfg_keys.*: recorded from upstream with this very file)."""
    from warpstr_amd import pore_model
    level = np.load(pore_model._DATA)
    i = np.arange(len(level))
    new = level * 1.04 + ((i * 2654435761) % 1009) / 400.0 - 1.2   # (products, sums and one exact division: no libm)
    with open(path, 'w') as f:
        f.write('kmer\tlevel_mean\tlevel_stdv\n')
        for k, v in enumerate(new):
            kmer = ''.join('ACGT'[(k >> (2 * (5 - b))) & 3] for b in range(6))
            f.write(f'{kmer}\t{float(v)!r}\t1.5\n')
    return path


_STAND_IN = {}


def _register_stand_in_filter(h, filter_id: int):
    """libhdf5 refuses to create a dataset whose pipeline names a filter it does not have; so the id is registered with a filter
    function that is never run (chunks are written and read already coded: H5Dwrite_chunk / H5Dread_chunk) and would pass the
    bytes through if it were."""
    import ctypes as C
    if filter_id in _STAND_IN:
        return
    fn_t = C.CFUNCTYPE(C.c_size_t, C.c_uint, C.c_size_t, C.POINTER(C.c_uint), C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_void_p))

    class H5ZClass2(C.Structure):
        _fields_ = [('version', C.c_int), ('id', C.c_int), ('encoder_present', C.c_uint), ('decoder_present', C.c_uint),
                    ('name', C.c_char_p), ('can_apply', C.c_void_p), ('set_local', C.c_void_p), ('filter', fn_t)]
    fn = fn_t(lambda flags, n_cd, cd, nbytes, buf_size, buf: nbytes)
    cls = H5ZClass2(1, filter_id, 1, 1, b'stand-in for the VBZ plugin (tests)', None, None, fn)
    h.H5Zregister.restype, h.H5Zregister.argtypes = C.c_int, [C.c_void_p]
    assert h.H5Zregister(C.byref(cls)) >= 0
    _STAND_IN[filter_id] = (cls, fn)   # (libhdf5 keeps the pointers)


def write_vbz_fast5(path: str, reads: dict, chunk_len: int, zigzag: bool = True, level: int = 1, skip_filter_on=()) -> str:
    """A multi-read .fast5 (`read_<id>/Raw/Signal`) whose int16 signals are stored as VBZ chunks (HDF5 filter 32020, version 0),
    written WITHOUT the filter plugin: the dataset is created with the filter in its pipeline (optional, so that libhdf5 accepts a
    filter it does not have) and every chunk is handed over already coded (H5Dwrite_chunk) -- u32 byte count, zstd frame (level !=
    0) around the StreamVByte block of oracle/vbz.py's encoder.  As HDF5 does for a real filter, a dataset's last chunk codes a
    WHOLE chunk (chunk_len samples, padded with zeros).  skip_filter_on: chunk numbers stored as plain samples with the chunk's
    filter mask set (what HDF5 does when an optional filter fails)."""
    import ctypes as C
    import struct

    from oracle import vbz
    from warpstr_amd import fast5
    h, zs = fast5._libs()
    hid = C.c_int64
    for fn, res, args in [('H5Fcreate', hid, [C.c_char_p, C.c_uint, hid, hid]), ('H5Gcreate2', hid, [hid, C.c_char_p, hid, hid, hid]),
                          ('H5Screate_simple', hid, [C.c_int, C.POINTER(C.c_uint64), C.c_void_p]),
                          ('H5Dcreate2', hid, [hid, C.c_char_p, hid, hid, hid, hid, hid]), ('H5Pcreate', hid, [hid]),
                          ('H5Pset_chunk', C.c_int, [hid, C.c_int, C.POINTER(C.c_uint64)]),
                          ('H5Pset_filter', C.c_int, [hid, C.c_int, C.c_uint, C.c_size_t, C.POINTER(C.c_uint)]),
                          ('H5Dwrite_chunk', C.c_int, [hid, hid, C.c_uint32, C.POINTER(C.c_uint64), C.c_size_t, C.c_void_p])]:
        f = getattr(h, fn)
        f.restype, f.argtypes = res, args
    _register_stand_in_filter(h, fast5.VBZ_FILTER)
    zs.ZSTD_compressBound.restype = C.c_size_t
    zs.ZSTD_compressBound.argtypes = [C.c_size_t]
    zs.ZSTD_compress.restype = C.c_size_t
    zs.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
    fid = h.H5Fcreate(path.encode(), 2, 0, 0)
    assert fid >= 0
    i16 = hid.in_dll(h, 'H5T_NATIVE_SHORT_g').value
    for rid, sig in reads.items():
        sig = np.ascontiguousarray(sig, dtype=np.int16)
        for g in (f'read_{rid}', f'read_{rid}/Raw'):
            h.H5Gclose(h.H5Gcreate2(fid, g.encode(), 0, 0, 0))
        # (extendible, as the files of the sequencer are: a chunk may then be longer than the dataset)
        sp = h.H5Screate_simple(1, (C.c_uint64 * 1)(len(sig)), C.cast((C.c_uint64 * 1)(2 ** 64 - 1), C.c_void_p))
        pl = h.H5Pcreate(hid.in_dll(h, 'H5P_CLS_DATASET_CREATE_ID_g').value)
        assert h.H5Pset_chunk(pl, 1, (C.c_uint64 * 1)(chunk_len)) >= 0
        assert h.H5Pset_filter(pl, fast5.VBZ_FILTER, 1, 4, (C.c_uint * 4)(0, 2, int(zigzag), int(level))) >= 0   # flags 1 = optional
        d = h.H5Dcreate2(fid, f'read_{rid}/Raw/Signal'.encode(), i16, sp, 0, pl, 0)
        assert d >= 0
        for k, start in enumerate(range(0, len(sig), chunk_len)):
            part = np.zeros(chunk_len, np.int16)
            part[:len(sig) - start] = sig[start:start + chunk_len]
            if k in skip_filter_on:
                buf, mask = part.tobytes(), 1
            else:
                block = vbz.svb_encode(vbz.values_from_samples(part, zigzag)).tobytes()
                if level:
                    out = C.create_string_buffer(zs.ZSTD_compressBound(len(block)))
                    m = zs.ZSTD_compress(out, len(out), block, len(block), int(level))
                    block = out.raw[:m]
                buf, mask = struct.pack('<I', 2 * chunk_len) + block, 0
            assert h.H5Dwrite_chunk(d, 0, mask, (C.c_uint64 * 1)(start), len(buf), buf) >= 0
        h.H5Dclose(d)
        h.H5Pclose(pl)
        h.H5Sclose(sp)
    h.H5Fclose(fid)
    return path


def free_port() -> int:
    """A TCP port for a rendezvous on 127.0.0.1, taken BELOW the kernel's range of ephemeral ports (32768-60999 by default): a port
    of that range found free by binding port 0 can be the source port of somebody's outgoing connection a moment later -- gloo
    opens many -- and the rendezvous then fails with EADDRINUSE (seen once in a round's last GPU run)."""
    import random
    import socket
    rng = random.Random(os.getpid() * 7919 + int.from_bytes(os.urandom(4), 'little'))
    for _ in range(200):
        port = rng.randrange(20000, 30000)
        s = socket.socket()
        try:
            s.bind(('127.0.0.1', port))
            return port
        except OSError:
            continue
        finally:
            s.close()
    raise RuntimeError('no free port between 20000 and 29999')
