"""Raw-signal reader for .fast5 files (the input of `Fast5.get_data_processed`, reference src/schemas/fast5.py:45-57).

The reference opens the file with h5py and relies on the VBZ HDF5 filter plugin (filter 32020) being installed.  Neither
is a dependency here: the container format is read through the system libhdf5 (ctypes), and VBZ chunks are fetched raw
(`H5Dread_chunk`) and decoded in-process -- zstd frame -> StreamVByte (2-bit length keys, little endian) -> zig-zag ->
running sum -- following the published VBZ version-0 layout.  Only the int16 `Signal` dataset is needed by the caller.

Layouts understood: single-read files written by the reference's steps 1-2 / `prepare_caller_only.py:8-22`
(`Raw/Reads/<first>/Signal`) and multi-read files (`read_<id>/Raw/Signal`).
"""
import ctypes as C
import ctypes.util
import glob
import os
import struct
from typing import List, Optional

import numpy as np

VBZ_FILTER = 32020
_hid = C.c_int64
_h5 = None
_zstd = None


class Fast5Error(RuntimeError):
    pass


def _find(name: str, extra: List[str]) -> Optional[str]:
    cand = ctypes.util.find_library(name)
    if cand:
        return cand
    for pat in extra:
        hits = sorted(glob.glob(pat))
        if hits:
            return hits[0]
    return None


def _libs():
    global _h5, _zstd
    if _h5 is not None:
        return _h5, _zstd
    p = os.environ.get('WARPSTR_LIBHDF5') or _find('hdf5', ['/opt/conda/lib/libhdf5.so*', '/usr/lib/*/libhdf5*.so*'])
    if not p:
        raise Fast5Error('libhdf5 not found (set WARPSTR_LIBHDF5 to its path)')
    h = C.CDLL(p)
    z = _find('zstd', ['/opt/conda/lib/libzstd.so*', '/usr/lib/*/libzstd.so*'])
    if not z:
        raise Fast5Error('libzstd not found')
    zs = C.CDLL(z)
    h.H5open()
    for fn, res, args in [
            ('H5Fopen', _hid, [C.c_char_p, C.c_uint, _hid]), ('H5Fclose', C.c_int, [_hid]),
            ('H5Gopen2', _hid, [_hid, C.c_char_p, _hid]), ('H5Gclose', C.c_int, [_hid]),
            ('H5Gget_num_objs', C.c_int, [_hid, C.POINTER(C.c_uint64)]),
            ('H5Gget_objname_by_idx', C.c_ssize_t, [_hid, C.c_uint64, C.c_char_p, C.c_size_t]),
            ('H5Lexists', C.c_int, [_hid, C.c_char_p, _hid]),
            ('H5Dopen2', _hid, [_hid, C.c_char_p, _hid]), ('H5Dclose', C.c_int, [_hid]),
            ('H5Dget_space', _hid, [_hid]), ('H5Sclose', C.c_int, [_hid]),
            ('H5Sget_simple_extent_npoints', C.c_int64, [_hid]),
            ('H5Dget_create_plist', _hid, [_hid]), ('H5Pclose', C.c_int, [_hid]),
            ('H5Pget_nfilters', C.c_int, [_hid]),
            ('H5Pget_filter2', C.c_int, [_hid, C.c_uint, C.POINTER(C.c_uint), C.POINTER(C.c_size_t),
                                         C.POINTER(C.c_uint), C.c_size_t, C.c_char_p, C.POINTER(C.c_uint)]),
            ('H5Pget_layout', C.c_int, [_hid]),
            ('H5Pget_chunk', C.c_int, [_hid, C.c_int, C.POINTER(C.c_uint64)]),
            ('H5Dget_chunk_storage_size', C.c_int, [_hid, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
            ('H5Dread_chunk', C.c_int, [_hid, _hid, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_void_p]),
            ('H5Dread', C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]),
            ('H5Eset_auto2', C.c_int, [_hid, C.c_void_p, C.c_void_p])]:
        f = getattr(h, fn)
        f.restype, f.argtypes = res, args
    h.H5Eset_auto2(0, None, None)  # errors are reported through return codes below, not printed by the library
    zs.ZSTD_getFrameContentSize.restype = C.c_uint64
    zs.ZSTD_getFrameContentSize.argtypes = [C.c_char_p, C.c_size_t]
    zs.ZSTD_decompress.restype = C.c_size_t
    zs.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t]
    zs.ZSTD_isError.restype = C.c_uint
    zs.ZSTD_isError.argtypes = [C.c_size_t]
    _h5, _zstd = h, zs
    return h, zs


def streamvbyte_decode(buf: np.ndarray, n: int) -> np.ndarray:
    """n uint32 values from a StreamVByte block: ceil(n/4) key bytes (2 bits per value = byte length - 1, first value in
    the low bits), then the little-endian value bytes back to back."""
    n_keys = (n + 3) // 4
    if buf.size < n_keys:
        raise Fast5Error('StreamVByte block shorter than its key area')
    keys = buf[:n_keys]
    lens = np.stack([(keys >> s) & 3 for s in (0, 2, 4, 6)], axis=1).reshape(-1)[:n].astype(np.int64) + 1
    ends = np.cumsum(lens)
    if n_keys + (int(ends[-1]) if n else 0) > buf.size:
        raise Fast5Error('StreamVByte block shorter than its keys say')
    starts = ends - lens + n_keys
    vals = np.zeros(n, dtype=np.uint32)
    for k in range(4):
        m = lens > k
        vals[m] |= buf[starts[m] + k].astype(np.uint32) << np.uint32(8 * k)
    return vals


_VBZ_C = False


def _vbz_c():
    """csrc/seam_helper.c's decoder loop (warpstr_amd/_seam_helper.so, loaded without the GIL), or None: the NumPy decoder
    below then does the same arithmetic.  Host-side file reading, not the caller's compute path."""
    global _VBZ_C
    if _VBZ_C is False:
        _VBZ_C = None
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_seam_helper.so')
        if os.path.exists(path) and not os.environ.get('WARPSTR_NO_SEAM_HELPER'):
            try:
                lib = C.CDLL(path)
                fn = lib.wsx_seam_vbz_decode_i16
                fn.restype = C.c_int64
                fn.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]
                _VBZ_C = fn
            except (OSError, AttributeError):
                _VBZ_C = None
    return _VBZ_C


def vbz_decode_chunk(chunk: bytes, int_size: int, zigzag: bool, version: int, zstd_level: int) -> np.ndarray:
    """One VBZ-filtered HDF5 chunk -> integer samples.  Layout: u32 uncompressed byte count, then (zstd_level != 0) a
    zstd frame holding the StreamVByte block of the delta-coded, optionally zig-zag-mapped, samples."""
    if version != 0:
        raise Fast5Error(f'VBZ version {version} is not supported (version 0 = 2-bit-key StreamVByte is)')
    if int_size not in (1, 2, 4):
        raise Fast5Error(f'VBZ integer size {int_size}')
    if len(chunk) < 4:
        raise Fast5Error('VBZ chunk too short')
    n_bytes, = struct.unpack_from('<I', chunk, 0)
    n = n_bytes // int_size
    body = chunk[4:]
    if zstd_level != 0:
        _, zs = _libs()
        size = zs.ZSTD_getFrameContentSize(body, len(body))
        if size >= (1 << 62):
            raise Fast5Error('VBZ chunk does not hold a sized zstd frame')
        out = C.create_string_buffer(max(int(size), 1))
        got = zs.ZSTD_decompress(out, size, body, len(body))
        if zs.ZSTD_isError(got) or got != size:
            raise Fast5Error('zstd decompression of a VBZ chunk failed')
        body = out.raw[:size]
    fn = _vbz_c() if int_size == 2 else None
    if fn is not None:
        block = np.frombuffer(body, dtype=np.uint8)
        out = np.empty(n, dtype=np.int16)
        rc = fn(block.ctypes.data_as(C.c_void_p), block.size, n, int(bool(zigzag)), out.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise Fast5Error('StreamVByte block shorter than its key area' if rc == -1 else 'StreamVByte block shorter than its keys say')
        return out
    vals = streamvbyte_decode(np.frombuffer(body, dtype=np.uint8), n)
    if zigzag:
        delta = (vals >> np.uint32(1)).astype(np.int32) ^ -(vals & np.uint32(1)).astype(np.int32)
    else:
        delta = vals.astype(np.int32)
    return np.cumsum(delta, dtype=np.int64).astype({1: np.int8, 2: np.int16, 4: np.int32}[int_size])


class Fast5File:
    """Read-only view of the raw signals of a .fast5 file."""

    def __init__(self, path: str):
        self.h, _ = _libs()
        self.path = path
        self.fid = self.h.H5Fopen(path.encode(), 0, 0)
        if self.fid < 0:
            raise Fast5Error(f'cannot open {path} as HDF5')

    def close(self):
        if self.fid >= 0:
            self.h.H5Fclose(self.fid)
            self.fid = -1

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def _children(self, group: str) -> List[str]:
        g = self.h.H5Gopen2(self.fid, group.encode(), 0)
        if g < 0:
            raise Fast5Error(f'{self.path}: no group {group}')
        n = C.c_uint64()
        self.h.H5Gget_num_objs(g, C.byref(n))
        buf = C.create_string_buffer(512)
        out = []
        for i in range(n.value):
            self.h.H5Gget_objname_by_idx(g, i, buf, 512)
            out.append(buf.value.decode())
        self.h.H5Gclose(g)
        return out

    def _exists(self, path: str) -> bool:
        cur = ''
        for part in path.strip('/').split('/'):
            cur = f'{cur}/{part}' if cur else part
            if self.h.H5Lexists(self.fid, cur.encode(), 0) <= 0:
                return False
        return True

    def read_ids(self) -> List[str]:
        """Read ids of a multi-read file ([] for a single-read file)."""
        return [k[5:] for k in self._children('/') if k.startswith('read_')]

    def signal_path(self, read_id: Optional[str] = None) -> str:
        if read_id is not None and self._exists(f'read_{read_id}/Raw/Signal'):
            return f'read_{read_id}/Raw/Signal'
        if self._exists('Raw/Reads'):  # single-read layout: first read, as fast5.py:50-52
            names = self._children('Raw/Reads')
            if names:
                return f'Raw/Reads/{names[0]}/Signal'
        if read_id is None:
            ids = self.read_ids()
            if len(ids) == 1:
                return f'read_{ids[0]}/Raw/Signal'
        raise Fast5Error(f'{self.path}: no raw signal' + (f' for read {read_id}' if read_id else ''))

    def _open_signal(self, read_id: Optional[str]):
        """(dataset id, samples, VBZ parameters [version, integer size, zig-zag, zstd level] or None, samples per chunk or 0)."""
        h = self.h
        # (a multi-read file's dataset is opened by its name straight away: probing the three levels of its path first costs as
        # much again on a file whose metadata is cold -- and every read of a run is read exactly once)
        d = h.H5Dopen2(self.fid, f'read_{read_id}/Raw/Signal'.encode(), 0) if read_id is not None else -1
        if d < 0:
            d = h.H5Dopen2(self.fid, self.signal_path(read_id).encode(), 0)
        if d < 0:
            raise Fast5Error(f'{self.path}: cannot open the signal dataset')
        try:
            sp = h.H5Dget_space(d)
            n = h.H5Sget_simple_extent_npoints(sp)
            h.H5Sclose(sp)
            pl = h.H5Dget_create_plist(d)
            vbz = None
            for i in range(max(h.H5Pget_nfilters(pl), 0)):
                flags, ne, cd, fc = C.c_uint(), C.c_size_t(8), (C.c_uint * 8)(), C.c_uint()
                name = C.create_string_buffer(64)
                if h.H5Pget_filter2(pl, i, C.byref(flags), C.byref(ne), cd, 64, name, C.byref(fc)) == VBZ_FILTER:
                    vbz = list(cd)[:ne.value] + [0] * 4
            chunk_len = (C.c_uint64 * 1)(0)
            chunked = h.H5Pget_layout(pl) == 2 and h.H5Pget_chunk(pl, 1, chunk_len) == 1
            h.H5Pclose(pl)
            if vbz is not None and not chunked:
                raise Fast5Error(f'{self.path}: VBZ filter on a dataset that is not chunked')
            return d, int(n), vbz, int(chunk_len[0]) if chunked else 0
        except Exception:
            h.H5Dclose(d)
            raise

    def signal_length(self, read_id: Optional[str] = None) -> int:
        """Samples of a read's raw signal (metadata only: nothing is decoded)."""
        d, n, _, _ = self._open_signal(read_id)
        self.h.H5Dclose(d)
        return n

    def raw_signal(self, read_id: Optional[str] = None) -> np.ndarray:
        """The DAC samples of a read (int16), whatever the dataset's storage filter."""
        return self.raw_signal_into(read_id, None)

    def raw_signal_into(self, read_id: Optional[str], out: Optional[np.ndarray]) -> np.ndarray:
        """raw_signal() decoded straight into `out` (a C-contiguous int16 array of exactly the read's length: a slice of a
        staging buffer, main_wrapper_loci's reader processes); out=None allocates; a callable is asked for the destination once the
        length is known (out(n) -> array of n samples)."""
        h = self.h
        d, n, vbz, chunk_len = self._open_signal(read_id)
        try:
            if out is None:
                out = np.empty(n, dtype=np.int16)
            elif callable(out):
                out = out(n)
            if out.dtype != np.int16 or out.ndim != 1 or out.size != n or not out.flags.c_contiguous or not out.flags.writeable:
                raise Fast5Error(f'{self.path}: the destination must be a writable contiguous int16 array of {n} samples')
            if vbz is None:  # contiguous / gzip / ...: the library's own pipeline handles it
                native_i16 = _hid.in_dll(h, 'H5T_NATIVE_SHORT_g').value
                if h.H5Dread(d, native_i16, 0, 0, 0, out.ctypes.data_as(C.c_void_p)) < 0:
                    raise Fast5Error(f'{self.path}: H5Dread failed')
                return out
            version, int_size, zigzag, level = vbz[0], vbz[1], vbz[2], vbz[3]
            native = _vbz_native() if (version == 0 and int_size == 2) else None
            done = 0
            for start in range(0, n, chunk_len):
                off, size, mask = (C.c_uint64 * 1)(start), C.c_uint64(), C.c_uint32()
                if h.H5Dget_chunk_storage_size(d, off, C.byref(size)) < 0 or size.value == 0:
                    raise Fast5Error(f'{self.path}: missing chunk at sample {start}')
                buf = _scratch(size.value)
                if h.H5Dread_chunk(d, 0, off, C.byref(mask), buf) < 0:
                    raise Fast5Error(f'{self.path}: H5Dread_chunk failed at sample {start}')
                want = min(chunk_len, n - start)
                if mask.value & 1:  # the filter was skipped when this chunk was written: plain samples
                    part = np.frombuffer(buf, dtype=np.int16, count=size.value // 2)[:want]
                    out[start:start + len(part)] = part
                    got = len(part)
                elif native is not None and want == min(chunk_len, struct.unpack_from('<I', buf, 0)[0] // 2):
                    fn, f_size, f_dec = native
                    got = fn(buf, size.value, int(bool(zigzag)), int(level), f_size, f_dec, out.ctypes.data + 2 * start, want)
                    if got < 0:
                        raise Fast5Error(f'{self.path}: ' + _VBZ_ERRORS.get(int(got), f'VBZ decoder error {got}'))
                else:
                    part = vbz_decode_chunk(bytes(buf[:size.value]), int_size, bool(zigzag), version, level)[:want]
                    if part.dtype != np.int16:
                        raise Fast5Error(f'{self.path}: decoded samples of {part.dtype}, the dataset holds int16')
                    out[start:start + len(part)] = part
                    got = len(part)
                done += int(got)
            if done != n:
                raise Fast5Error(f'{self.path}: decoded {done} samples, the dataset holds {n} int16')
            return out
        finally:
            h.H5Dclose(d)


_SCRATCH = None
_VBZ_NATIVE = False
_VBZ_ERRORS = {-1: 'VBZ chunk too short', -2: 'VBZ chunk does not hold a sized zstd frame', -3: 'zstd decompression of a VBZ chunk failed',
               -4: 'StreamVByte block shorter than its key area', -5: 'StreamVByte block shorter than its keys say',
               -6: 'a VBZ chunk holds more samples than the dataset says'}


def _scratch(nbytes: int):
    """A buffer of at least nbytes that lives with the process (a chunk's compressed bytes: no allocation per read)."""
    global _SCRATCH
    if _SCRATCH is None or len(_SCRATCH) < nbytes:
        _SCRATCH = (C.c_char * max(nbytes + nbytes // 2, 1 << 20))()
    return _SCRATCH


def _vbz_native():
    """(wsh_vbz_decode_i16 of warpstr_amd/_host_loci.so, ZSTD_getFrameContentSize, ZSTD_decompress as addresses): one call without
    the GIL per chunk -- zstd frame -> StreamVByte -> zig-zag -> running sum -> the destination; or None (library not built: the
    decoders above do the same arithmetic)."""
    global _VBZ_NATIVE
    if _VBZ_NATIVE is False:
        _VBZ_NATIVE = None
        from . import _hostlib
        lib = _hostlib.lib()
        if lib is not None and hasattr(lib, 'wsh_vbz_decode_i16'):
            _, zs = _libs()
            fn = lib.wsh_vbz_decode_i16
            fn.restype = C.c_int64
            fn.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
            _VBZ_NATIVE = (fn, C.cast(zs.ZSTD_getFrameContentSize, C.c_void_p).value, C.cast(zs.ZSTD_decompress, C.c_void_p).value)
    return _VBZ_NATIVE


def read_raw_signal(path: str, read_id: Optional[str] = None) -> np.ndarray:
    with Fast5File(path) as f:
        return f.raw_signal(read_id)
