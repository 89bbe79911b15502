"""Raw-signal reader for .fast5 files (the input of `Fast5.get_data_processed`, reference src/schemas/fast5.py:45-57).

The reference opens the file with h5py and relies on the VBZ HDF5 filter plugin (filter 32020) being installed.  Neither
is a dependency here: the container format is read through the system libhdf5 (ctypes), and VBZ chunks are fetched raw
(`H5Dread_chunk`) and decoded in-process -- zstd frame -> StreamVByte (2-bit length keys, little endian) -> zig-zag ->
running sum -- following the published VBZ version-0 layout.  Only the int16 `Signal` dataset is needed by the caller.

Layouts understood: single-read files written by the reference's steps 1-2 / `prepare_caller_only.py:8-22`
(`Raw/Reads/<first>/Signal`) and multi-read files (`read_<id>/Raw/Signal`).
"""
import ctypes as C
import os
import struct
from typing import Optional

import numpy as np

from ._h5core import VBZ_FILTER, Fast5Core, Fast5Error, NeedsNumpy  # noqa: F401
from ._h5core import libs as _libs, vbz_native as _vbz_native  # noqa: F401 -- the names this module always had


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


class Fast5File(Fast5Core):
    """Read-only view of the raw signals of a .fast5 file (the container and the VBZ fast path: _h5core.Fast5Core)."""

    def raw_signal(self, read_id: Optional[str] = None) -> np.ndarray:
        """The DAC samples of a read (int16), whatever the dataset's storage filter."""
        return self.raw_signal_into(read_id, None)

    def raw_signal_into(self, read_id: Optional[str], out) -> np.ndarray:
        """raw_signal() decoded straight into `out` (a C-contiguous int16 array of exactly the read's length: a slice of a
        staging buffer, main_wrapper_loci's reader processes); out=None allocates; a callable is asked for the destination once the
        length is known (out(n) -> array of n samples)."""
        dest = []

        def place(n):
            o = np.empty(n, dtype=np.int16) if out is None else (out(n) if callable(out) else out)
            if o.dtype != np.int16 or o.ndim != 1 or o.size != n or not o.flags.c_contiguous or not o.flags.writeable:
                raise Fast5Error(f'{self.path}: the destination must be a writable contiguous int16 array of {n} samples')
            dest.append(o)
            return o.ctypes.data
        try:
            self.decode_to(read_id, place)
            return dest[0]
        except NeedsNumpy:   # another integer size, a chunk longer than its share of the dataset, _host_loci.so not built
            pass
        d, n, vbz, chunk_len = self._open_signal(read_id)
        try:
            if not dest:
                place(n)
            o = dest[0]
            version, int_size, zigzag, level = vbz[0], vbz[1], vbz[2], vbz[3]
            done = 0
            for start, want, buf, size, plain in self._chunks(d, n, chunk_len):
                if plain:  # the filter was skipped when this chunk was written: plain samples
                    part = np.frombuffer(buf, dtype=np.int16, count=size // 2)[:want]
                else:
                    part = vbz_decode_chunk(bytes(buf[:size]), int_size, bool(zigzag), version, level)[:want]
                    if part.dtype != np.int16:
                        raise Fast5Error(f'{self.path}: decoded samples of {part.dtype}, the dataset holds int16')
                o[start:start + len(part)] = part
                done += len(part)
            if done != n:
                raise Fast5Error(f'{self.path}: decoded {done} samples, the dataset holds {n} int16')
            return o
        finally:
            self.h.H5Dclose(d)


def read_raw_signal(path: str, read_id: Optional[str] = None) -> np.ndarray:
    with Fast5File(path) as f:
        return f.raw_signal(read_id)
