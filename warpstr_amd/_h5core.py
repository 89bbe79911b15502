"""The part of the .fast5 reader (warpstr_amd/fast5.py) that needs neither NumPy nor the package: the system libhdf5 and libzstd
through ctypes, a file's raw-signal datasets, and a read decoded to a memory ADDRESS (VBZ chunks through wsh_vbz_decode_i16 of
warpstr_amd/_host_loci.so).  A reader process of main_wrapper_loci imports this and nothing heavier: sixteen interpreters that start
together come up in ~80 ms this way, in ~0.7 s when each imports NumPy (scripts/exp_startup_probe.py) -- and for that long they
take the CPUs from the parent's set-up.

Reference: what `Fast5.get_data_processed` reads (src/schemas/fast5.py:45-57); layouts as in fast5.py's header."""
import ctypes as C
import glob
import os
from typing import List, Optional

VBZ_FILTER = 32020
_hid = C.c_int64
_h5 = None
_zstd = None
_paths = None


class Fast5Error(RuntimeError):
    pass


class NeedsNumpy(Exception):
    """A chunk this module does not decode (VBZ of another integer size, or _host_loci.so not built): fast5.py's decoders do."""


def _find(name: str, extra: List[str]) -> Optional[str]:
    import ctypes.util
    cand = ctypes.util.find_library(name)
    if cand:
        return cand
    for pat in extra:
        hits = sorted(glob.glob(pat))
        if hits:
            return hits[0]
    return None


def lib_paths():
    """(libhdf5, libzstd) as this process loads them.  WARPSTR_LIBHDF5 / WARPSTR_LIBZSTD name them outright (the parent of reader
    processes sets both: looking a library up costs each of them tens of milliseconds and several child processes)."""
    global _paths
    if _paths is None:
        p = os.environ.get('WARPSTR_LIBHDF5') or _find('hdf5', ['/opt/conda/lib/libhdf5.so*', '/usr/lib/*/libhdf5*.so*'])
        if not p:
            raise Fast5Error('libhdf5 not found (set WARPSTR_LIBHDF5 to its path)')
        z = os.environ.get('WARPSTR_LIBZSTD') or _find('zstd', ['/opt/conda/lib/libzstd.so*', '/usr/lib/*/libzstd.so*'])
        if not z:
            raise Fast5Error('libzstd not found (set WARPSTR_LIBZSTD to its path)')
        _paths = (p, z)
    return _paths


def libs():
    global _h5, _zstd
    if _h5 is not None:
        return _h5, _zstd
    p, z = lib_paths()
    h = C.CDLL(p)
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


_SCRATCH = None
_VBZ_NATIVE = False
_VBZ_CONTEXT: list = []   # [(wsh_vbz_context, ZSTD_createDCtx, address of ZSTD_decompressDCtx)] once vbz_native() has bound them
_THREAD = None            # threading.local(): has this thread its decompression context?
PLAIN, SVB_ZIGZAG, SVB = 0, 1, 2   # what a block written by Fast5Core.blocks_to holds (= WSX_VBZ_* of include/warpstr_hip.h)
ZSTD_SVB_ZIGZAG, ZSTD_SVB = 3, 4   # ... or the zstd frame around such a block, left for wsx_zstd_decode
VBZ_ERRORS = {-1: 'VBZ chunk too short', -2: 'VBZ chunk does not hold a sized zstd frame', -3: 'zstd decompression of a VBZ chunk failed',
              -4: 'StreamVByte block shorter than its key area', -5: 'StreamVByte block shorter than its keys say',
              -6: 'a VBZ block does not fit the room made for it'}



def scratch(nbytes: int):
    """A buffer of at least nbytes that lives with the process (a chunk's compressed bytes: no allocation per read)."""
    global _SCRATCH
    if _SCRATCH is None or len(_SCRATCH) < nbytes:
        _SCRATCH = (C.c_char * max(nbytes + nbytes // 2, 1 << 20))()
    return _SCRATCH


def vbz_native():
    """(wsh_vbz_decode_i16 of warpstr_amd/_host_loci.so, ZSTD_getFrameContentSize, ZSTD_decompress as addresses): one call without
    the GIL per chunk -- zstd frame -> StreamVByte -> zig-zag -> running sum -> the destination; or None (library not built, or
    WARPSTR_NO_HOST_NATIVE: fast5.py's decoders do the same arithmetic)."""
    global _VBZ_NATIVE
    if _VBZ_NATIVE is False:
        _VBZ_NATIVE = None
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_host_loci.so')
        if os.path.exists(path) and not os.environ.get('WARPSTR_NO_HOST_NATIVE'):
            try:
                lib = C.CDLL(path)
                fn, unpack = lib.wsh_vbz_decode_i16, lib.wsh_vbz_unpack
            except (OSError, AttributeError):
                return None
            _, zs = libs()
            fn.restype = C.c_int64
            fn.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
            unpack.restype = C.c_int64
            unpack.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
            _VBZ_NATIVE = (fn, C.cast(zs.ZSTD_getFrameContentSize, C.c_void_p).value, C.cast(zs.ZSTD_decompress, C.c_void_p).value, unpack)
            try:
                zs.ZSTD_createDCtx.restype = C.c_void_p
                lib.wsh_vbz_context.restype, lib.wsh_vbz_context.argtypes = None, [C.c_void_p, C.c_void_p]
                _VBZ_CONTEXT.append((lib.wsh_vbz_context, zs.ZSTD_createDCtx, C.cast(zs.ZSTD_decompressDCtx, C.c_void_p).value))
            except AttributeError:
                pass
    return _VBZ_NATIVE


def thread_context():
    """A zstd decompression context of its own for the calling thread, kept for the life of the process (the native decoders
    then skip the context ZSTD_decompress makes and drops per call: 9 % of a frame's time)."""
    global _THREAD
    if _THREAD is None:
        import threading
        _THREAD = threading.local()
    if getattr(_THREAD, 'done', False) or not _VBZ_CONTEXT:
        return
    _THREAD.done = True
    set_context, create, decompress = _VBZ_CONTEXT[0]
    dctx = create()
    if dctx:
        set_context(dctx, decompress)


class Fast5Core:
    """Read-only view of the raw-signal datasets of a .fast5 file."""

    def __init__(self, path: str):
        self.h, _ = libs()
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

    def _chunks(self, d, n: int, chunk_len: int):
        """(first sample, samples wanted, the stored bytes in scratch(), their count, filter-skipped?) of each chunk of an open
        dataset."""
        h = self.h
        for start in range(0, n, chunk_len):
            off, size, mask = (C.c_uint64 * 1)(start), C.c_uint64(), C.c_uint32()
            if h.H5Dget_chunk_storage_size(d, off, C.byref(size)) < 0 or size.value == 0:
                raise Fast5Error(f'{self.path}: missing chunk at sample {start}')
            buf = scratch(size.value)
            if h.H5Dread_chunk(d, 0, off, C.byref(mask), buf) < 0:
                raise Fast5Error(f'{self.path}: H5Dread_chunk failed at sample {start}')
            yield start, min(chunk_len, n - start), buf, size.value, bool(mask.value & 1)

    def decode_to(self, read_id: Optional[str], place) -> int:
        """The DAC samples of a read (int16) written at the address place(n) returns once the length n is known; returns n.
        NeedsNumpy for a read only fast5.py's decoders read (place() may have been asked by then: the same destination is theirs)."""
        h = self.h
        d, n, vbz, chunk_len = self._open_signal(read_id)
        try:
            native = None
            if vbz is not None:
                native = vbz_native() if (vbz[0] == 0 and vbz[1] == 2) else None
                if native is None:
                    raise NeedsNumpy()
            addr = int(place(n))
            if vbz is None:  # contiguous / gzip / ...: the library's own pipeline handles it
                native_i16 = _hid.in_dll(h, 'H5T_NATIVE_SHORT_g').value
                if h.H5Dread(d, native_i16, 0, 0, 0, C.c_void_p(addr)) < 0:
                    raise Fast5Error(f'{self.path}: H5Dread failed')
                return n
            fn, f_size, f_dec = native[:3]
            thread_context()
            zigzag, level = int(bool(vbz[2])), int(vbz[3])
            done = 0
            for start, want, buf, size, plain in self._chunks(d, n, chunk_len):
                if plain:  # the filter was skipped when this chunk was written: plain samples
                    got = min(size // 2, want)
                    C.memmove(addr + 2 * start, buf, 2 * got)
                else:
                    if size < 4:
                        raise Fast5Error(f'{self.path}: ' + VBZ_ERRORS[-1])
                    # (a dataset's last chunk codes chunk-length samples -- HDF5 hands a filter whole chunks --: the first `want`)
                    got = fn(buf, size, zigzag, level, f_size, f_dec, addr + 2 * start, want)
                    if got < 0:
                        raise Fast5Error(f'{self.path}: ' + VBZ_ERRORS.get(int(got), f'VBZ decoder error {got}'))
                done += int(got)
            if done != n:
                raise Fast5Error(f'{self.path}: decoded {done} samples, the dataset holds {n} int16')
            return n
        finally:
            h.H5Dclose(d)

    @staticmethod
    def frame_for_device(frame: bytes):
        """The content size a zstd frame declares if the device decoder takes the frame (csrc/wsx_zstd.hip: no dictionary, a declared
        size, at most 32 blocks, none of a reserved type, no "treeless" literals before a block has brought a Huffman tree), else None --
        from its headers alone."""
        b, n = frame, len(frame)
        if n < 6 or b[:4] != b'\x28\xb5\x2f\xfd':
            return None
        fhd = b[4]
        flag, single, did = fhd >> 6, (fhd >> 5) & 1, fhd & 3
        if did or fhd & 8:
            return None
        pos = 5 + (0 if single else 1)
        fcs = (1 if single else 0) if flag == 0 else (2, 4, 8)[flag - 1]
        if fcs == 0 or pos + fcs > n:
            return None
        content = int.from_bytes(b[pos:pos + fcs], 'little') + (256 if fcs == 2 else 0)
        pos += fcs
        if content > 32 << 17:
            return None
        tree = False
        for _ in range(32):
            if pos + 3 > n:
                return None
            bh = int.from_bytes(b[pos:pos + 3], 'little')
            pos += 3
            last, btype, size = bh & 1, (bh >> 1) & 3, bh >> 3
            if btype == 3 or pos + (1 if btype == 1 else size) > n:
                return None
            if btype == 2:
                if size < 1 or ((b[pos] & 3) == 3 and not tree):
                    return None
                tree = tree or (b[pos] & 3) == 2
            pos += 1 if btype == 1 else size
            if last:
                return content
        return None

    def blocks_to(self, read_id: Optional[str], place, device_zstd: bool = False):
        """A read as the device decoder takes it (wsx_vbz_decode): every chunk of a VBZ dataset as the StreamVByte block inside
        its zstd frame -- zstd is the part a GPU does not do --, anything else as plain int16 samples.  place(nbytes) -> the address
        the next block goes to (asked once per block, in order).  Returns (samples of the read, [(kind, bytes, samples wanted, values
        coded, content bytes), ...]); device_zstd: a chunk whose zstd frame the device can decode is left as that frame (kind ZSTD_*,
        content bytes = what the frame declares; 0 for the other kinds).
        NeedsNumpy as for decode_to (before place() is asked for the first time, or after: the caller starts the read again)."""
        h = self.h
        d, n, vbz, chunk_len = self._open_signal(read_id)
        try:
            if vbz is None:  # contiguous / gzip / ...: the library's own pipeline, plain samples
                native_i16 = _hid.in_dll(h, 'H5T_NATIVE_SHORT_g').value
                if h.H5Dread(d, native_i16, 0, 0, 0, C.c_void_p(int(place(2 * n)))) < 0:
                    raise Fast5Error(f'{self.path}: H5Dread failed')
                return n, [(PLAIN, 2 * n, n, n, 0)]
            native = vbz_native() if (vbz[0] == 0 and vbz[1] == 2) else None
            if native is None:
                raise NeedsNumpy()
            _, f_size, f_dec, unpack = native
            thread_context()
            _, zs = libs()
            kind, level = (SVB_ZIGZAG if vbz[2] else SVB), int(vbz[3])
            blocks, done, n_out = [], 0, C.c_int64()
            for start, want, buf, size, plain in self._chunks(d, n, chunk_len):
                if plain:  # the filter was skipped when this chunk was written: plain samples
                    got = min(size // 2, want)
                    C.memmove(int(place(2 * got)), buf, 2 * got)
                    blocks.append((PLAIN, 2 * got, got, got, 0))
                else:
                    if size < 4:
                        raise Fast5Error(f'{self.path}: ' + VBZ_ERRORS[-1])
                    if device_zstd and level != 0:
                        content = self.frame_for_device(bytes(buf[4:size]))
                        if content is not None and content <= 5 * max(chunk_len, want) + 64:
                            C.memmove(int(place(size - 4)), C.addressof(buf) + 4, size - 4)
                            coded = int.from_bytes(bytes(buf[:4]), 'little') // 2
                            got = min(want, coded)
                            blocks.append((kind + 2, size - 4, got, coded, content))
                            done += got
                            continue
                    if level != 0:
                        cap = zs.ZSTD_getFrameContentSize(C.cast(C.addressof(buf) + 4, C.c_char_p), size - 4)
                        # (a block of n values holds ceil(n/4) key bytes and at most 4 n value bytes: a frame that declares more is
                        # corrupt -- and would otherwise size an arena)
                        if cap >= (1 << 62) or cap > 5 * max(chunk_len, want) + 64:
                            raise Fast5Error(f'{self.path}: ' + VBZ_ERRORS[-2])
                    else:
                        cap = size - 4
                    nb = unpack(buf, size, level, f_size, f_dec, int(place(int(cap))), int(cap), C.byref(n_out))
                    if nb < 0:
                        raise Fast5Error(f'{self.path}: ' + VBZ_ERRORS.get(int(nb), f'VBZ decoder error {nb}'))
                    got = min(want, int(n_out.value))
                    blocks.append((kind, int(nb), got, int(n_out.value), 0))
                done += got
            if done != n:
                raise Fast5Error(f'{self.path}: decoded {done} samples, the dataset holds {n} int16')
            return n, blocks
        finally:
            h.H5Dclose(d)
