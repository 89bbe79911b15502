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
// vbz_decode_kernel: a workgroup of 256 lanes per block (a read's chunk: 60-170 k samples), 2048 values per round -- a lane takes
// two key bytes 256 apart, of each its four values: their lengths, a workgroup scan, <= 16 data bytes, four differences, a second workgroup scan,
// four samples; the next round's keys and the 4 KB window its value bytes lie in are fetched a round ahead (the window into LDS).  Integer work, bit-exact against the host decoders (warpstr_amd/fast5.py, csrc/host_loci.cpp).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <type_traits>

#include "../../include/warpstr_hip.h"

namespace {

constexpr int VBZ_LANES = 256;
constexpr int VBZ_TILES = 2;               // key bytes a lane takes per round, 256 key bytes apart: two independent chains between
                                           // the same two barriers (1: 0.72 ms per 2 048 blocks, 2: 0.665, 4: 0.686)
constexpr int VBZ_WINDOW = VBZ_TILES * VBZ_LANES * 16; // bytes of values a round can ask for (four bytes each)

__device__ __forceinline__ int wave_inclusive_sum(int v)
{
    int r = v;
    r += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);  // row_shr:1
    r += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);  // row_shr:2
    r += __builtin_amdgcn_update_dpp(0, v, 0x113, 0xf, 0xf, true);  // row_shr:3
    r += __builtin_amdgcn_update_dpp(0, r, 0x114, 0xf, 0xe, true);  // row_shr:4, banks 1-3
    r += __builtin_amdgcn_update_dpp(0, r, 0x118, 0xf, 0xc, true);  // row_shr:8, banks 2-3
    r += __builtin_amdgcn_update_dpp(0, r, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1 and 3
    r += __builtin_amdgcn_update_dpp(0, r, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2 and 3
    return r;
}

typedef uint32_t vbz_u32x4 __attribute__((ext_vector_type(4)));

// 16-byte chunk `chunk` of the window that starts at the 16-byte boundary at or below `from`: one wide load where that lies
// inside [lo, hi) (the whole of src), byte by byte at the two ends of the buffer
__device__ __forceinline__ vbz_u32x4 window_load(const uint8_t *from, const uint8_t *lo, const uint8_t *hi, int chunk)
{
    // (the address as an offset from `lo`, the kernel's own argument: a GLOBAL load.  A pointer made from an integer is a flat one, and
    // a flat load counts as an LDS operation too: the loads that are meant to be in flight during a round were waited for at the round's
    // first wait for LDS)
    const long long rel = (from - lo) - (long long)((uintptr_t)from & 15) + 16 * chunk, n = hi - lo;
    vbz_u32x4 w = {0, 0, 0, 0};
    if (rel >= 0 && rel + 16 <= n) {
        w = *(const vbz_u32x4 *)(lo + rel);
    } else if (rel + 16 > 0 && rel < n) {
        uint32_t t[4] = {0, 0, 0, 0};
        for (int i = 0; i < 16; i++)
            if (rel + i >= 0 && rel + i < n) t[i >> 2] |= (uint32_t)lo[rel + i] << (8 * (i & 3));
        w = vbz_u32x4{t[0], t[1], t[2], t[3]};
    }
    return w;
}

__global__ __launch_bounds__(VBZ_LANES) void vbz_decode_kernel(const uint8_t *__restrict__ src, long long src_total,
                                                               const wsx_vbz_block *__restrict__ blocks, int16_t *__restrict__ dst,
                                                               int32_t *__restrict__ status)
{
    constexpr int T = VBZ_TILES, W = VBZ_LANES / 64;
    __shared__ int wsum[2][T][W];                                                 // per tile and wavefront: bytes, differences
    __shared__ __attribute__((aligned(16))) uint8_t win[2][VBZ_WINDOW + 16];      // the value bytes of this round and of the next
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
    const int nkeys = (n + 3) >> 2;                       // key bytes that matter: those of the samples wanted
    const int key_area = (B.n_values + 3) >> 2;           // ... of all the values the block codes
    const uint8_t *data = p + key_area;
    const int data_bytes = (int)(B.src_bytes - key_area); // (>= n_values and < 2^31: checked on the host)
    int doff = 0; // bytes of the values before this round (the same in every lane; 32 bits: 64-bit offsets cost two instructions each)
    int acc = 0;        // sum of the differences before this round; only its low 16 bits matter
    bool bad = false;
    // A round's chain would be: key byte (global) -> scan -> value bytes (global, at an address the scan gives) -> scan -> store.
    // Both loads are taken out of it: the keys of round r+1 depend on nothing and are fetched in round r; the value bytes of round
    // r+1 start where round r's end -- known after round r's FIRST scan -- so the window that holds them, whatever the keys of
    // round r+1 will say, is fetched then (a 16-byte load per lane and tile), lands in LDS at the end of round r and is read from
    // there.  A lane takes T key bytes per round, one of each tile of 256: their chains are independent, the barriers shared.
    int key_next[T];
#pragma unroll
    for (int t = 0; t < T; t++) key_next[t] = t * VBZ_LANES + tid < nkeys ? p[t * VBZ_LANES + tid] : 0;
    // (the window is T * 4 096 + 16 bytes: it starts at a 16-byte boundary up to 15 bytes before the round's first value byte; the
    // last chunk is lane 0's extra load)
#pragma unroll
    for (int t = 0; t < T; t++) *(vbz_u32x4 *)&win[0][16 * (t * VBZ_LANES + tid)] = window_load(data, src, src + src_total, t * VBZ_LANES + tid);
    if (tid == 0) *(vbz_u32x4 *)&win[0][VBZ_WINDOW] = window_load(data, src, src + src_total, T * VBZ_LANES);
    int par = 0;
    // (a round all of whose values exist -- every round but a block's last -- is compiled without the tests for that)
    auto round = [&](const int k0, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        int ki[T], valid[T], len[T][4], tl[T], incl[T];
#pragma unroll
        for (int t = 0; t < T; t++) {
            ki[t] = k0 + t * VBZ_LANES + tid;
            valid[t] = FULL ? 4 : (ki[t] < nkeys ? min(4, n - 4 * ki[t]) : 0); // values of this key byte that exist
            const int key = key_next[t];
            key_next[t] = ki[t] + T * VBZ_LANES < nkeys ? p[ki[t] + T * VBZ_LANES] : 0;
#pragma unroll
            for (int j = 0; j < 4; j++) len[t][j] = valid[t] > j ? ((key >> (2 * j)) & 3) + 1 : 0;
            tl[t] = len[t][0] + len[t][1] + len[t][2] + len[t][3];
            incl[t] = wave_inclusive_sum(tl[t]);
            if (lane == 63) wsum[0][t][wave] = incl[t];
        }
        __syncthreads(); // (also: the window of this round, written at the end of the last, is in LDS)
        int before[T], round_bytes = 0;
#pragma unroll
        for (int t = 0; t < T; t++) {
            before[t] = round_bytes; // the tiles before this one ...
#pragma unroll
            for (int w = 0; w < W; w++) {
                const int s = wsum[0][t][w];
                before[t] += w < wave ? s : 0; // ... and the wavefronts before this one in it
                round_bytes += s;
            }
        }
        vbz_u32x4 next[T], next_tail = {0, 0, 0, 0}; // (in flight during this round)
#pragma unroll
        for (int t = 0; t < T; t++) next[t] = window_load(data + doff + round_bytes, src, src + src_total, t * VBZ_LANES + tid);
        if (tid == 0) next_tail = window_load(data + doff + round_bytes, src, src + src_total, T * VBZ_LANES);
        const int align = (int)((uintptr_t)(data + doff) & 15);
        int sum[T][4], total[T], vincl[T];
#pragma unroll
        for (int t = 0; t < T; t++) {
            const int rel = before[t] + incl[t] - tl[t]; // this key byte's first value byte, from the round's first
            uint32_t v[4] = {0, 0, 0, 0};
            // (the bytes a value does not have are not read: all sixteen read unconditionally and cut afterwards was slower)
            if (rel + tl[t] <= data_bytes - doff) {
                const uint8_t *q = &win[par][align + rel];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    uint32_t x = 0;
                    if (len[t][j] > 0) x = q[0];
                    if (len[t][j] > 1) x |= (uint32_t)q[1] << 8;
                    if (len[t][j] > 2) x |= (uint32_t)q[2] << 16;
                    if (len[t][j] > 3) x |= (uint32_t)q[3] << 24;
                    v[j] = x;
                    q += len[t][j];
                }
            } else if (tl[t] > 0) {
                bad = true; // the keys ask for bytes the block does not have: zeros from here on
            }
            int run = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                run += zigzag ? (int)((v[j] >> 1) ^ (0u - (v[j] & 1u))) : (int)v[j];
                sum[t][j] = run;
            }
            total[t] = run;
            vincl[t] = wave_inclusive_sum(run);
            if (lane == 63) wsum[1][t][wave] = vincl[t];
        }
#pragma unroll
        for (int t = 0; t < T; t++) *(vbz_u32x4 *)&win[par ^ 1][16 * (t * VBZ_LANES + tid)] = next[t]; // (last read a round ago, two barriers back)
        if (tid == 0) *(vbz_u32x4 *)&win[par ^ 1][VBZ_WINDOW] = next_tail;
        __syncthreads();
        int round_sum = 0;
#pragma unroll
        for (int t = 0; t < T; t++) {
            int vbefore = round_sum;
#pragma unroll
            for (int w = 0; w < W; w++) {
                const int s = wsum[1][t][w];
                vbefore += w < wave ? s : 0;
                round_sum += s;
            }
            const int base = acc + vbefore + vincl[t] - total[t];
            int16_t *o = out + 4 * (size_t)ki[t];
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (valid[t] > j) o[j] = (int16_t)(base + sum[t][j]);
        }
        acc += round_sum;
        doff += round_bytes;
        // (wsum[0] is written again after the second barrier of this round, wsum[1] after the first of the next: two suffice)
    };
    for (int k0 = 0; k0 < nkeys; k0 += T * VBZ_LANES, par ^= 1) {
        if (4 * (k0 + T * VBZ_LANES) <= n) round(k0, std::true_type{});
        else round(k0, std::false_type{});
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
        if (B.src_bytes > 0x7fffffff) {
            wsx_internal_set_error("wsx_vbz_decode: a block of 2 GB or more");
            return WSX_ERR_INVALID;
        }
        if (B.n_values < B.n_samples || B.reserved != 0) {
            wsx_internal_set_error("wsx_vbz_decode: a block must code at least the samples wanted of it (n_values >= n_samples, reserved = 0)");
            return WSX_ERR_INVALID;
        }
        const int64_t nv = B.n_values;
        const int64_t least = B.kind == WSX_VBZ_PLAIN ? 2 * n : (nv + 3) / 4 + nv; // key area + a byte per value
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
    hipLaunchKernelGGL(vbz_decode_kernel, dim3((unsigned)n_blocks), dim3(VBZ_LANES), 0, st, src, (long long)src_bytes, (const wsx_vbz_block *)d, dst, status);
    VCHK(hipGetLastError());
    return WSX_SUCCESS;
} catch (...) {
    return wsx_internal_on_exception();
}
