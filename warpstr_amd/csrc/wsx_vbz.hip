// wsx_vbz.hip -- the samples of VBZ-compressed signal datasets on the GPU (what stands between the .fast5 file and the
// signal loader of wsx_prep.hip).
//
// Upstream reads `Raw/Signal` through h5py (Fast5.get_data_processed, src/schemas/fast5.py:50-52) and leaves the decoding to the
// HDF5 filter plugin (filter 32020, ont-vbz-hdf-plugin; not in the upstream tree).  Its published version-0 layout for 2-byte
// integers: u32 byte count, then a zstd frame holding a StreamVByte block -- ceil(n/4) key bytes (two bits per value: byte
// length - 1, first value in the low bits), then the values' little-endian bytes back to back; the values are the zig-zag
// mapped differences of consecutive samples.  zstd is undone on the host (warpstr_amd/_h5core.py: a reader process writes the
// frame's content into its page-locked arena); the rest is two prefix sums, done here:
//   where the bytes of a value start  = sum of the byte lengths before it      (scan over the keys)
//   the sample                        = sum of the differences up to it, mod 2^16 (scan over the values)
// vbz_decode_kernel: a workgroup of 256 lanes per block (a read's chunk: 60-170 k samples), 1024 values per round -- a lane takes
// a key byte, i.e. four values: their lengths, a workgroup scan, <= 16 data bytes, four differences, a second workgroup scan,
// four samples; the next round's keys are fetched a round ahead.  Integer work, bit-exact against the host decoders (warpstr_amd/fast5.py, csrc/host_loci.cpp).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "../../include/warpstr_hip.h"

namespace {

constexpr int VBZ_LANES = 256;
constexpr int VBZ_KEYS = 1; // key bytes per lane and round (two -- eight values, half the barriers -- is slower: 1.00 vs 0.84 ms per
                            // 2 048 blocks, the lane's chain of dependent byte loads is twice as long)

__device__ __forceinline__ int wave_inclusive_sum(int v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

__global__ __launch_bounds__(VBZ_LANES) void vbz_decode_kernel(const uint8_t *__restrict__ src, const wsx_vbz_block *__restrict__ blocks,
                                                               int16_t *__restrict__ dst, int32_t *__restrict__ status)
{
    __shared__ int wsum[2][VBZ_LANES / 64]; // per wavefront: bytes, differences
    const wsx_vbz_block B = blocks[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = B.n_samples;
    int16_t *out = dst + B.dst_offset;
    const uint8_t *p = src + B.src_offset;
    if (B.kind == WSX_VBZ_PLAIN) { // (a dataset without the filter, or a chunk written with the filter skipped)
        for (int i = tid; i < n; i += VBZ_LANES) out[i] = (int16_t)((uint32_t)p[2 * (size_t)i] | ((uint32_t)p[2 * (size_t)i + 1] << 8));
        return;
    }
    const bool zigzag = B.kind == WSX_VBZ_SVB_ZIGZAG;
    const int nkeys = (n + 3) >> 2;
    const uint8_t *data = p + nkeys;
    const long long data_bytes = B.src_bytes - nkeys; // (>= n: checked on the host)
    long long doff = 0; // bytes of the values before this round (the same in every lane)
    int acc = 0;        // sum of the differences before this round; only its low 16 bits matter
    bool bad = false;
    constexpr int V = 4 * VBZ_KEYS;
    // the keys of the next round are fetched a round ahead: they depend on nothing (the values' bytes do: on the scan)
    int key_next[VBZ_KEYS];
#pragma unroll
    for (int q = 0; q < VBZ_KEYS; q++) {
        const int ki = VBZ_KEYS * tid + q;
        key_next[q] = ki < nkeys ? p[ki] : 0;
    }
    for (int k0 = 0; k0 < nkeys; k0 += VBZ_KEYS * VBZ_LANES) {
        const int kfirst = k0 + VBZ_KEYS * tid; // this lane's first key byte; its values: 4 * kfirst ..
        int len[V];
#pragma unroll
        for (int q = 0; q < VBZ_KEYS; q++) {
            const int key = key_next[q];
            const int kn = kfirst + VBZ_KEYS * VBZ_LANES + q;
            key_next[q] = kn < nkeys ? p[kn] : 0;
#pragma unroll
            for (int j = 0; j < 4; j++) len[4 * q + j] = 4 * (kfirst + q) + j < n ? ((key >> (2 * j)) & 3) + 1 : 0;
        }
        int tl = 0;
#pragma unroll
        for (int j = 0; j < V; j++) tl += len[j];
        const int incl = wave_inclusive_sum(tl);
        if (lane == 63) wsum[0][wave] = incl;
        __syncthreads();
        int before = 0, round_bytes = 0;
#pragma unroll
        for (int w = 0; w < VBZ_LANES / 64; w++) {
            const int s = wsum[0][w];
            before += w < wave ? s : 0;
            round_bytes += s;
        }
        const long long my = doff + before + incl - tl;
        int sum[V]; // running sum of this lane's differences
        int run = 0;
        if (my + tl <= data_bytes) {
            const uint8_t *q = data + my;
#pragma unroll
            for (int j = 0; j < V; j++) {
                uint32_t x = 0;
                if (len[j] > 0) x = q[0];
                if (len[j] > 1) x |= (uint32_t)q[1] << 8;
                if (len[j] > 2) x |= (uint32_t)q[2] << 16;
                if (len[j] > 3) x |= (uint32_t)q[3] << 24;
                q += len[j];
                run += zigzag ? (int)((x >> 1) ^ (0u - (x & 1u))) : (int)x;
                sum[j] = run;
            }
        } else {
            if (tl > 0) bad = true; // the keys ask for bytes the block does not have: zeros from here on
#pragma unroll
            for (int j = 0; j < V; j++) sum[j] = 0;
        }
        const int vincl = wave_inclusive_sum(run);
        if (lane == 63) wsum[1][wave] = vincl;
        __syncthreads();
        int vbefore = 0, round_sum = 0;
#pragma unroll
        for (int w = 0; w < VBZ_LANES / 64; w++) {
            const int s = wsum[1][w];
            vbefore += w < wave ? s : 0;
            round_sum += s;
        }
        const int base = acc + vbefore + vincl - run;
        int16_t *o = out + 4 * (size_t)kfirst;
#pragma unroll
        for (int j = 0; j < V; j++)
            if (len[j] > 0) o[j] = (int16_t)(base + sum[j]);
        acc += round_sum;
        doff += round_bytes;
        // (wsum[0] is written again after the second barrier of this round, wsum[1] after the first of the next: two suffice)
    }
    if (bad && status) status[blockIdx.x] = 1;
}

} // namespace

int wsx_internal_device(wsx_caller *c);
hipStream_t wsx_internal_stream(wsx_caller *c);
void wsx_internal_set_error(const char *msg);
extern "C" int wsx_internal_on_exception(void);
// a slot of the handle's ring for this function's block table: pinned host memory and device memory of >= bytes each, and the
// event recorded after the slot's last upload
hipError_t wsx_internal_vbz_slot(wsx_caller *c, size_t bytes, void **host, void **dev, hipEvent_t *last_use);

#define VCHK(expr)                                                                                       \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) {                                                                          \
            char b_[512];                                                                                \
            snprintf(b_, sizeof(b_), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            wsx_internal_set_error(b_);                                                                  \
            return WSX_ERR_HIP;                                                                          \
        }                                                                                                \
    } while (0)

extern "C" int wsx_vbz_decode(wsx_caller *c, const uint8_t *src, int64_t src_bytes, const wsx_vbz_block *blocks, int64_t n_blocks,
                              int16_t *dst, int64_t dst_samples, int32_t *status)
try {
    if (!c || n_blocks < 0 || src_bytes < 0 || dst_samples < 0 || (n_blocks > 0 && (!blocks || !src || !dst))) {
        wsx_internal_set_error("wsx_vbz_decode: null or negative argument");
        return WSX_ERR_INVALID;
    }
    if (n_blocks > 0x7fffffff) {
        wsx_internal_set_error("wsx_vbz_decode: more than 2^31 - 1 blocks in one call");
        return WSX_ERR_INVALID;
    }
    for (int64_t b = 0; b < n_blocks; b++) {
        const wsx_vbz_block &B = blocks[b];
        const int64_t n = B.n_samples;
        if (B.kind < WSX_VBZ_PLAIN || B.kind > WSX_VBZ_SVB || n < 0 || B.src_offset < 0 || B.src_bytes < 0 || B.dst_offset < 0 ||
            B.src_bytes > src_bytes || B.src_offset > src_bytes - B.src_bytes || n > dst_samples || B.dst_offset > dst_samples - n) {
            wsx_internal_set_error("wsx_vbz_decode: a block lies outside src or dst");
            return WSX_ERR_INVALID;
        }
        const int64_t least = B.kind == WSX_VBZ_PLAIN ? 2 * n : (n + 3) / 4 + n; // key area + a byte per value
        if (B.src_bytes < least) {
            wsx_internal_set_error(B.kind == WSX_VBZ_PLAIN ? "wsx_vbz_decode: a block of plain samples shorter than 2 bytes per sample"
                                                           : "wsx_vbz_decode: StreamVByte block shorter than its key area and a byte per value");
            return WSX_ERR_INVALID;
        }
    }
    if (n_blocks == 0) return WSX_SUCCESS;
    VCHK(hipSetDevice(wsx_internal_device(c)));
    hipStream_t st = wsx_internal_stream(c);
    void *h = nullptr, *d = nullptr;
    hipEvent_t ev = nullptr;
    const size_t bytes = (size_t)n_blocks * sizeof(wsx_vbz_block);
    VCHK(wsx_internal_vbz_slot(c, bytes, &h, &d, &ev));
    VCHK(hipEventSynchronize(ev)); // the upload of three calls ago has left the slot (a no-op before that)
    memcpy(h, blocks, bytes);
    VCHK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st));
    VCHK(hipEventRecord(ev, st));
    if (status) VCHK(hipMemsetAsync(status, 0, (size_t)n_blocks * sizeof(int32_t), st));
    hipLaunchKernelGGL(vbz_decode_kernel, dim3((unsigned)n_blocks), dim3(VBZ_LANES), 0, st, src, (const wsx_vbz_block *)d, dst, status);
    VCHK(hipGetLastError());
    return WSX_SUCCESS;
} catch (...) {
    return wsx_internal_on_exception();
}
