// seam_helper.c -- the Python side of CallerWrapper.run(List[ReadSignal]) (upstream src/caller/wrapper.py:104-120) spends
// its time walking 50 000 Python objects.  These two loops do that walk with the CPython C API (loaded through
// ctypes.PyDLL, so the GIL is held): no arithmetic happens here, the HIP library gets the same pointers as before.
// Built by warpstr_amd/build.py into warpstr_amd/_seam_helper.so; caller.py falls back to its own loops if it is absent.
#include <Python.h>
#include <stdint.h>
#include <string.h>

// For every item of `workload`: ptrs[i] / lens[i] = address and length of item.<sig_attr> (a C-contiguous 1-d float64
// buffer), aut[i] = 1 if item.<rev_attr> is true else 0.  Returns n, or -(i+1) if item i does not qualify (the caller
// then converts the arrays itself), or INT64_MIN on a Python error.
// `keep` (a list, or NULL): every signal object is appended to it, so that the addresses stay valid for as long as the caller
// holds the list even if item.<sig_attr> is a property that builds a new array per access.
int64_t wsx_seam_collect(PyObject *workload, const char *sig_attr, const char *rev_attr, uintptr_t *ptrs, int64_t *lens,
                         int32_t *aut, PyObject *keep)
{
    PyObject *fast = PySequence_Fast(workload, "workload must be a sequence");
    if (!fast) return INT64_MIN;
    PyObject *sig_name = PyUnicode_InternFromString(sig_attr), *rev_name = PyUnicode_InternFromString(rev_attr);
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
    int64_t rc = n;
    for (Py_ssize_t i = 0; i < n && rc == n; i++) {
        PyObject *item = PySequence_Fast_GET_ITEM(fast, i);
        PyObject *sig = PyObject_GetAttr(item, sig_name);
        PyObject *rev = sig ? PyObject_GetAttr(item, rev_name) : NULL;
        if (!sig || !rev) {
            rc = INT64_MIN;
        } else {
            Py_buffer view;
            if (PyObject_GetBuffer(sig, &view, PyBUF_FORMAT | PyBUF_ND | PyBUF_C_CONTIGUOUS) != 0) {
                PyErr_Clear();
                rc = -(int64_t)(i + 1);
            } else {
                const char *f = view.format ? view.format : "B";
                if (*f == '@' || *f == '=' || *f == '<') f++;
                if (view.ndim != 1 || view.itemsize != 8 || strcmp(f, "d") != 0) {
                    rc = -(int64_t)(i + 1);
                } else {
                    ptrs[i] = (uintptr_t)view.buf;
                    lens[i] = (int64_t)view.shape[0];
                    if (keep && keep != Py_None && PyList_Append(keep, sig) != 0) rc = INT64_MIN;
                }
                PyBuffer_Release(&view);
            }
            const int t = PyObject_IsTrue(rev);
            if (t < 0) rc = INT64_MIN;
            aut[i] = t > 0;
        }
        Py_XDECREF(sig);
        Py_XDECREF(rev);
    }
    Py_XDECREF(sig_name);
    Py_XDECREF(rev_name);
    Py_DECREF(fast);
    return rc;
}

// Packs the called sequences of a batch: read r's bytes src[offsets[r] .. offsets[r] + len[r]) go to dst[pos[r] ..) with
// pos = exclusive prefix sum of len (written to pos[0..n]); len is read with a byte stride (a field of the result
// records).  dst must hold sum(len) bytes; returns that sum.
int64_t wsx_seam_pack_sequences(const uint8_t *src, const int64_t *offsets, const void *len_field, int64_t len_stride,
                                int64_t n, uint8_t *dst, int64_t *pos)
{
    int64_t total = 0;
    for (int64_t r = 0; r < n; r++) {
        int32_t len;
        memcpy(&len, (const char *)len_field + r * len_stride, sizeof len);
        if (len < 0) len = 0;
        pos[r] = total;
        if (dst) memcpy(dst + total, src + offsets[r], (size_t)len);
        total += len;
    }
    pos[n] = total;
    return total;
}

// For every item of `arrays` (a sequence of C-contiguous 1-d int16 buffers: the raw reads of a batch): ptrs[i] / lens[i] = address
// and number of samples.  Returns n, -(i+1) if item i is something else (the caller then lets NumPy convert), INT64_MIN on a
// Python error.  The buffers are only read while the caller keeps `arrays` alive.
int64_t wsx_seam_buffers_i16(PyObject *arrays, uintptr_t *ptrs, int64_t *lens)
{
    PyObject *fast = PySequence_Fast(arrays, "the reads must be a sequence");
    if (!fast) return INT64_MIN;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
    int64_t rc = n;
    for (Py_ssize_t i = 0; i < n && rc == n; i++) {
        Py_buffer view;
        if (PyObject_GetBuffer(PySequence_Fast_GET_ITEM(fast, i), &view, PyBUF_FORMAT | PyBUF_ND | PyBUF_C_CONTIGUOUS) != 0) {
            PyErr_Clear();
            rc = -(int64_t)(i + 1);
            break;
        }
        const char *f = view.format ? view.format : "B";
        if (*f == '@' || *f == '=' || *f == '<') f++;
        if (view.ndim != 1 || view.itemsize != 2 || strcmp(f, "h") != 0) rc = -(int64_t)(i + 1);
        else {
            ptrs[i] = (uintptr_t)view.buf;
            lens[i] = (int64_t)view.shape[0];
        }
        PyBuffer_Release(&view);
    }
    Py_DECREF(fast);
    return rc;
}

// ---- VBZ payloads (warpstr_amd/fast5.py) ----------------------------------------------------------------------------
// One StreamVByte block (2-bit length keys, ceil(n/4) key bytes first, then the little-endian value bytes back to back) of
// delta-coded, optionally zig-zag-mapped 16-bit samples -> the samples (running sum, wrapped to int16 as NumPy's cast
// does).  Plain C, no Python objects: fast5.py calls it through ctypes.CDLL, i.e. without the GIL.  The NumPy decoder in
// fast5.py is the same arithmetic (and the checker in tests/test_host_logic.py); this loop is ~40x faster on a 130 k-sample read.
// Returns 0, -1 if the block is shorter than its key area, -2 if shorter than its keys say.
int64_t wsx_seam_vbz_decode_i16(const uint8_t *svb, int64_t n_bytes, int64_t n, int32_t zigzag, int16_t *out)
{
    const int64_t n_keys = (n + 3) / 4;
    if (n_bytes < n_keys) return -1;
    const uint8_t *data = svb + n_keys, *end = svb + n_bytes;
    int64_t acc = 0;
    for (int64_t i = 0; i < n; i++) {
        const int len = ((svb[i >> 2] >> ((i & 3) * 2)) & 3) + 1;
        if (data + len > end) return -2;
        uint32_t v = 0;
        for (int k = 0; k < len; k++) v |= (uint32_t)data[k] << (8 * k);
        data += len;
        const int32_t delta = zigzag ? (int32_t)(v >> 1) ^ -(int32_t)(v & 1u) : (int32_t)v;
        acc += delta;
        out[i] = (int16_t)(uint16_t)(uint64_t)acc;
    }
    return 0;
}

// ---- multi-GPU host path (warpstr_amd/dist.py) ------------------------------------------------------------------------
// Longest-processing-time partition: items in the given order (descending work), each to the rank with the smallest load so
// far, the LOWEST rank among equals -- the same rule, and the same floating-point sums in the same order, as the NumPy loop
// in dist.shard_reads, so every rank derives the same partition whichever of the two it runs.  owner[order[i]] = rank.
void wsx_seam_lpt(const double *work, const int64_t *order, int64_t n, int32_t world, int64_t *owner)
{
    double load[1024];
    if (world > 1024) world = 1024;
    for (int r = 0; r < world; r++) load[r] = 0.0;
    for (int64_t i = 0; i < n; i++) {
        int best = 0;
        for (int r = 1; r < world; r++)
            if (load[r] < load[best]) best = r;
        owner[order[i]] = best;
        load[best] += work[order[i]];
    }
}

// Ragged pieces laid end to end: dst[total ..) = src[start[k] .. start[k] + len[k]) for k = 0..n-1.  Returns the bytes written.
int64_t wsx_seam_gather_pieces(const uint8_t *src, const int64_t *start, const int64_t *len, int64_t n, uint8_t *dst)
{
    int64_t total = 0;
    for (int64_t k = 0; k < n; k++) {
        if (len[k] > 0) memcpy(dst + total, src + start[k], (size_t)len[k]);
        total += len[k] > 0 ? len[k] : 0;
    }
    return total;
}

// ... and back: the k-th piece of src (pieces end to end, lengths len[idx[k]]) goes to dst[start[idx[k]] ..).  Returns the
// bytes read.
int64_t wsx_seam_scatter_pieces(const uint8_t *src, const int64_t *idx, int64_t n, const int64_t *start, const int64_t *len,
                                uint8_t *dst)
{
    int64_t at = 0;
    for (int64_t k = 0; k < n; k++) {
        const int64_t i = idx[k];
        if (len[i] > 0) memcpy(dst + start[i], src + at, (size_t)len[i]);
        at += len[i] > 0 ? len[i] : 0;
    }
    return at;
}
