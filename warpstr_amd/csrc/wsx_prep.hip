// wsx_prep.hip -- raw squiggle -> normalised squiggle segment on the GPU (the loader that feeds the caller).
//
// Upstream: Fast5.get_data_processed (src/schemas/fast5.py:45-57) =
//   remove_spikes 'Brute' (brute_remove, 90-101): samples > 1000 or < 250 at index > 2 are replaced, in index order
//       and in the raw integer dtype, by np.median(out[i-2:i+3]) (already-replaced neighbours are used);
//   normalize_signal_mad (104-114) over the WHOLE read: shift = mean(percentile(x, [46.5, 53.5])),
//       scale = median(|x - shift|), norm = (x - shift) / scale;
//   slice [l_start_raw : r_end_raw + 1].
// NumPy semantics restated: percentile 'linear' (virtual index (n-1)q, lerp a + (b-a)g, or b - (b-a)(1-g) for g >= 0.5),
// median = mean of the middle one/two order statistics, float -> int16 store truncates toward zero.
//
// Parallelisation (the three passes over the raw read are HBM streams: eight samples per 16-byte load)
//   copy_kernel       block per 4096 samples: raw -> working copy, smallest / largest value of the read, and the read's
//                     OUTLIER LIST (collected per block in LDS, appended with one atomic per block)
//   spike_list_kernel block per read, thread per listed outlier: the outlier at the head of a chain (gaps <= 2) repairs
//                     the whole chain serially -- chains further apart than 2 samples never touch each other's windows
//   spike_kernel      fallback for reads whose list overflowed (more than one outlier in 16 samples): every sample again
//   zero_kernel / hist_kernel   histogram per read over its occupied value range (int16 has only 2^16 values, so every
//                     order statistic comes from counting; no sort): a private LDS histogram per 16384-sample block,
//                     flushed bin by bin
//   stats_kernel      wavefront per read: cumulative walk -> percentiles -> shift; two-sided merge around shift
//                     -> median absolute deviation -> scale
//   norm_kernel       thread per output sample: (x - shift) / scale of the requested slice
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/warpstr_hip.h"
#include "wsx_device.h"

namespace {

struct PrepArgs {
    const int16_t *raw;
    int16_t *clean;
    const int64_t *roff;   // device, [n+1], offsets of the chunk's reads into raw/clean (already chunk-relative)
    const int64_t *seg_lo; // device, [n] l_start_raw
    const int64_t *seg_hi; // device, [n] r_end_raw (inclusive)
    const int64_t *ooff;   // device, [n+1] output offsets (chunk-relative)
    uint32_t *hist;        // [n][65536]; bin = value - (smallest value of the read); only the occupied range is touched
    int32_t *mm;           // [n][2]: smallest and largest raw value of the read
    double *shift_scale;   // [n][2]
    double *out;
    int n;
    int max_len;
    int vec; // raw and clean are 16-byte aligned: the streaming passes load eight samples per lane
    // outliers found by the copy pass (spike removal): per read a list of sample indices (segment of read r: entries
    // roff[r]/16 + 64 r .., capacity len/16 + 64) and a counter; a read whose list overflows is scanned sample by sample
    long long *ol_list;
    unsigned int *ol_count; // [n]
    uint8_t *done;          // [n]: 1 = the read went through short_read_kernel, the general kernels skip it
    int group;              // reads a block of the general kernels looks after (1; 256 for chunks short_read_kernel has been through)
};

// The general kernels, a block per read -- or, for a chunk short_read_kernel has been through, a block per `group` reads:
// nearly all of them are done then, and finding that out a block per read cost 0.07-0.23 ms per kernel and 100 k reads
// (0.85 ms of the 2.6 ms the loader added to a step).  The block looks at its reads' flags together and leaves if none is
// left; what is left (a read that spans more than PREP_SHORT_BINS values) is done read by read.  Inside the braces `r` is the
// read; `continue` leaves it (block-uniform conditions only: there are barriers inside).
#define PREP_READS_BEGIN(a, r)                                                                                      \
    const int prep_g_ = (a).group > 1 ? (a).group : 1;                                                              \
    const int prep_r0_ = (int)blockIdx.x * prep_g_;                                                                 \
    if (prep_g_ > 1) {                                                                                              \
        int left_ = 0;                                                                                              \
        for (int q_ = threadIdx.x; q_ < prep_g_ && prep_r0_ + q_ < (a).n; q_ += blockDim.x) left_ |= !(a).done[prep_r0_ + q_]; \
        if (!__syncthreads_or(left_)) return;                                                                       \
    }                                                                                                               \
    for (int r = prep_r0_; r < prep_r0_ + prep_g_ && r < (a).n; r++) {                                              \
        if (prep_g_ > 1 && r > prep_r0_) __syncthreads();                                                           \
        if ((a).done[r]) continue;
#define PREP_READS_END }

__device__ __forceinline__ long long ol_base(const PrepArgs &a, int r) { return a.roff[r] / 16 + 64ll * r; }
__device__ __forceinline__ unsigned int ol_cap(const PrepArgs &a, int r) { return (unsigned int)((a.roff[r + 1] - a.roff[r]) / 16 + 64); }
#define OL_OVERFLOW 0x80000000u

__device__ __forceinline__ bool is_outlier(const int16_t *raw, long long i)
{
    const int v = raw[i];
    return i > 2 && (v > 1000 || v < 250);
}

// The streaming passes move eight samples (16 bytes) per lane and load: groups of eight are aligned in BUFFER
// coordinates (the staging buffers are 256-byte aligned; a caller's device buffer is checked by the launcher), so a
// read that starts anywhere has a partial first and last group -- `first`/`count` say which of the eight belong to the
// piece [s0, s1) being processed.  vec = 0 falls back to one sample per lane.
typedef short short8 __attribute__((ext_vector_type(8)));

template <class Fn>
__device__ __forceinline__ void for_each_sample8(const int16_t *buf, long long s0, long long s1, int tid, int nthreads, const Fn &fn)
{
    for (long long g = (s0 >> 3) + tid; (g << 3) < s1; g += nthreads) {
        const short8 v = *(const short8 *)(buf + (g << 3));
        const long long base = g << 3;
        const int first = base < s0 ? (int)(s0 - base) : 0;
        const int last = base + 8 > s1 ? (int)(s1 - base) : 8; // exclusive
        fn(v, base, first, last);
    }
}

#define PREP_CHUNK 16384 // samples per block in the histogram pass (one private LDS histogram per block)
#define PREP_STREAM 4096 // samples per block in the copy and spike passes (latency-bound: many small blocks)

// copy + smallest / largest value of the read (spike removal writes medians of neighbours, so the range stays valid)
__global__ __launch_bounds__(256) void copy_kernel(PrepArgs a)
{
    __shared__ long long bl_list[256];
    __shared__ unsigned int bl_count, bl_base;
    PREP_READS_BEGIN(a, r)
    const long long len = a.roff[r + 1] - a.roff[r];
    const long long c0 = (long long)blockIdx.y * PREP_STREAM, c1 = c0 + PREP_STREAM < len ? c0 + PREP_STREAM : len;
    int lo = 32767, hi = -32768;
    // the block's outliers collect in LDS (bl_list, bl_count) and join the read's list with one atomic per block
    if (threadIdx.x == 0) bl_count = 0;
    __syncthreads();
    auto note_outlier = [&](long long i) {
        const unsigned int q = atomicAdd(&bl_count, 1u);
        if (q < 256) bl_list[q] = i;
    };
    if (a.vec) {
        for_each_sample8(a.raw, a.roff[r] + c0, a.roff[r] + c1, threadIdx.x, 256, [&](short8 v, long long base, int first, int last) {
            if (first == 0 && last == 8) {
                *(short8 *)(a.clean + base) = v;
            } else {
                for (int e = first; e < last; e++) a.clean[base + e] = v[e];
            }
#pragma unroll
            for (int e = 0; e < 8; e++)
                if (e >= first && e < last) {
                    lo = min(lo, (int)v[e]);
                    hi = max(hi, (int)v[e]);
                    const long long i = base + e - a.roff[r];
                    if (a.ol_list && i > 2 && (v[e] > 1000 || v[e] < 250)) note_outlier(i); // is_outlier: rare
                }
        });
    } else {
        for (long long i = c0 + threadIdx.x; i < c1; i += 256) {
            const int16_t v = a.raw[a.roff[r] + i];
            a.clean[a.roff[r] + i] = v;
            lo = min(lo, (int)v);
            hi = max(hi, (int)v);
            if (a.ol_list && i > 2 && (v > 1000 || v < 250)) note_outlier(i);
        }
    }
    for (int s = 32; s >= 1; s >>= 1) {
        lo = min(lo, __shfl_xor(lo, s));
        hi = max(hi, __shfl_xor(hi, s));
    }
    if ((threadIdx.x & 63) == 0 && c0 < c1) {
        atomicMin(&a.mm[2 * r], lo);
        atomicMax(&a.mm[2 * r + 1], hi);
    }
    if (a.ol_list) {
        __syncthreads();
        const unsigned int nb = bl_count;
        if (nb == 0) continue;
        if (threadIdx.x == 0) // more than 256 outliers in one block: the read is scanned sample by sample instead
            bl_base = nb > 256 ? atomicOr(&a.ol_count[r], OL_OVERFLOW) : atomicAdd(&a.ol_count[r], nb);
        __syncthreads();
        if (nb <= 256 && bl_base + nb <= ol_cap(a, r))
            for (unsigned int q = threadIdx.x; q < nb; q += 256) a.ol_list[ol_base(a, r) + bl_base + q] = bl_list[q];
    }
    PREP_READS_END
}

// spike_removal median3 / median5 (remove_spikes, src/schemas/fast5.py:68-75): scipy.signal.medfilt of the raw integer read
// with a window of W samples -- the middle order statistic of the window, the read padded with ZEROS beyond both ends --
// instead of copy_kernel: raw -> filtered working copy, and the smallest / largest FILTERED value (the padding can bring a
// zero into a read that has none).  A thread per sample; the W loads of neighbouring lanes overlap in L1.
__device__ __forceinline__ int med3(int a, int b, int c) { return max(min(a, b), min(max(a, b), c)); }

template <int W>
__global__ __launch_bounds__(256) void medfilt_kernel(PrepArgs a)
{
    static_assert(W == 3 || W == 5, "median3 / median5");
    PREP_READS_BEGIN(a, r)
    const long long len = a.roff[r + 1] - a.roff[r];
    const long long c0 = (long long)blockIdx.y * PREP_STREAM, c1 = c0 + PREP_STREAM < len ? c0 + PREP_STREAM : len;
    const int16_t *src = a.raw + a.roff[r];
    int lo = 32767, hi = -32768;
    for (long long i = c0 + threadIdx.x; i < c1; i += 256) {
        int v[W];
#pragma unroll
        for (int e = 0; e < W; e++) {
            const long long q = i - W / 2 + e;
            v[e] = (q >= 0 && q < len) ? (int)src[q] : 0;
        }
        int m;
        if constexpr (W == 3) {
            m = med3(v[0], v[1], v[2]);
        } else { // median of five: drop the smaller of the two pair minima and the larger of the two pair maxima
            const int a0 = min(v[0], v[1]), a1 = max(v[0], v[1]), b0 = min(v[2], v[3]), b1 = max(v[2], v[3]);
            m = med3(max(a0, b0), min(a1, b1), v[4]);
        }
        a.clean[a.roff[r] + i] = (int16_t)m;
        lo = min(lo, m);
        hi = max(hi, m);
    }
    for (int s = 32; s >= 1; s >>= 1) {
        lo = min(lo, __shfl_xor(lo, s));
        hi = max(hi, __shfl_xor(hi, s));
    }
    if ((threadIdx.x & 63) == 0 && c0 < c1) {
        atomicMin(&a.mm[2 * r], lo);
        atomicMax(&a.mm[2 * r + 1], hi);
    }
    PREP_READS_END
}

// zeroes the occupied range of every read's histogram
__global__ __launch_bounds__(256) void zero_kernel(PrepArgs a)
{
    PREP_READS_BEGIN(a, r)
    const int range = a.mm[2 * r + 1] - a.mm[2 * r] + 1;
    uint32_t *h = a.hist + (size_t)r * 65536;
    for (int b = blockIdx.y * 256 + threadIdx.x; b < range; b += gridDim.y * 256) h[b] = 0u;
    PREP_READS_END
}

// np.median of 2..5 int16 values, stored back into int16 (truncation toward zero)
__device__ int16_t median_small(int *w, int cnt)
{
    for (int x = 1; x < cnt; x++) { // insertion sort
        const int key = w[x];
        int y = x - 1;
        while (y >= 0 && w[y] > key) {
            w[y + 1] = w[y];
            y--;
        }
        w[y + 1] = key;
    }
    double med;
    if (cnt & 1) med = (0.0 + (double)w[cnt / 2]) / 1.0;
    else med = ((0.0 + (double)w[cnt / 2 - 1]) + (double)w[cnt / 2]) / 2.0;
    return (int16_t)med;
}

__global__ __launch_bounds__(256) void spike_kernel(PrepArgs a)
{
    PREP_READS_BEGIN(a, r)
    const long long len = a.roff[r + 1] - a.roff[r];
    const int16_t *raw = a.raw + a.roff[r];
    int16_t *out = a.clean + a.roff[r];
    const long long c0 = (long long)blockIdx.y * PREP_STREAM, c1 = c0 + PREP_STREAM < len ? c0 + PREP_STREAM : len;
    auto fix_chain_from = [&](long long i) { // i is an outlier: if it heads a chain, fix the whole chain serially
        if ((i >= 1 && is_outlier(raw, i - 1)) || (i >= 2 && is_outlier(raw, i - 2))) return;
        long long j = i;
        while (true) {
            int w[5];
            int cnt = 0;
            for (long long q = j - 2; q < j + 3 && q < len; q++) w[cnt++] = out[q];
            out[j] = median_small(w, cnt);
            if (j + 1 < len && is_outlier(raw, j + 1)) j = j + 1;
            else if (j + 2 < len && is_outlier(raw, j + 2)) j = j + 2;
            else break;
        }
    };
    if (a.ol_list && a.ol_count[r] <= ol_cap(a, r)) continue; // the listed outliers are handled by spike_list_kernel
    if (a.vec) {
        const long long ro = a.roff[r];
        for_each_sample8(a.raw, ro + c0, ro + c1, threadIdx.x, 256, [&](short8 v, long long base, int first, int last) {
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const long long i = base + e - ro; // index in the read
                if (e >= first && e < last && i > 2 && (v[e] > 1000 || v[e] < 250)) fix_chain_from(i);
            }
        });
    } else {
        for (long long i = c0 + threadIdx.x; i < c1; i += 256)
            if (is_outlier(raw, i)) fix_chain_from(i);
    }
    PREP_READS_END
}

// Spike removal from the copy pass's outlier lists: one block per read, one thread per listed outlier; the one that heads
// a chain (neither of the two samples before it is an outlier) repairs the whole chain serially, exactly as spike_kernel.
__global__ __launch_bounds__(256) void spike_list_kernel(PrepArgs a)
{
    PREP_READS_BEGIN(a, r)
    const unsigned int total = a.ol_count[r];
    if (total > ol_cap(a, r)) continue; // overflow: spike_kernel scans this read
    const long long len = a.roff[r + 1] - a.roff[r];
    const int16_t *raw = a.raw + a.roff[r];
    int16_t *out = a.clean + a.roff[r];
    const long long *list = a.ol_list + ol_base(a, r);
    for (unsigned int t = threadIdx.x; t < total; t += 256) {
        const long long i = list[t];
        if ((i >= 1 && is_outlier(raw, i - 1)) || (i >= 2 && is_outlier(raw, i - 2))) continue;
        long long j = i;
        while (true) {
            int w[5];
            int cnt = 0;
            for (long long q = j - 2; q < j + 3 && q < len; q++) w[cnt++] = out[q];
            out[j] = median_small(w, cnt);
            if (j + 1 < len && is_outlier(raw, j + 1)) j = j + 1;
            else if (j + 2 < len && is_outlier(raw, j + 2)) j = j + 2;
            else break;
        }
    }
    PREP_READS_END
}

#define PREP_LDS_BINS 8192
// Counts one PREP_CHUNK of a read.  A raw read occupies a few thousand of the 65536 possible values, so the block counts
// into a private LDS histogram over [smallest value, +PREP_LDS_BINS) and adds only its non-zero bins to the read's
// histogram in HBM (a global atomic per SAMPLE serialises on the few hundred hot bins: it was half of the loader's
// time); reads with a wider range count straight into HBM.
__global__ __launch_bounds__(256) void hist_kernel(PrepArgs a)
{
    // (dynamic: PREP_LDS_BINS counters, or none when the block looks after a group of reads -- a block that asks for 32 KB of
    // LDS waits for a CU with that much free, beside the fills 0.2 ms per launch, only to find its reads done; the odd read
    // that is left then counts straight into HBM)
    extern __shared__ uint32_t lh[];
    PREP_READS_BEGIN(a, r)
    const long long len = a.roff[r + 1] - a.roff[r];
    const long long c0 = (long long)blockIdx.y * PREP_CHUNK, c1 = c0 + PREP_CHUNK < len ? c0 + PREP_CHUNK : len;
    if (c0 >= c1) continue;
    const int16_t *x = a.clean + a.roff[r];
    const int vmin = a.mm[2 * r], range = a.mm[2 * r + 1] - vmin + 1;
    uint32_t *h = a.hist + (size_t)r * 65536;
    if (range <= PREP_LDS_BINS && a.group <= 1) {
        for (int b = threadIdx.x; b < range; b += 256) lh[b] = 0u;
        __syncthreads();
        if (a.vec) {
            for_each_sample8(a.clean, a.roff[r] + c0, a.roff[r] + c1, threadIdx.x, 256, [&](short8 v, long long, int first, int last) {
#pragma unroll
                for (int e = 0; e < 8; e++)
                    if (e >= first && e < last) atomicAdd(&lh[(int)v[e] - vmin], 1u);
            });
        } else {
            for (long long i = c0 + threadIdx.x; i < c1; i += 256) atomicAdd(&lh[(int)x[i] - vmin], 1u);
        }
        __syncthreads();
        for (int b = threadIdx.x; b < range; b += 256) {
            const uint32_t c = lh[b];
            if (c) atomicAdd(&h[b], c);
        }
    } else {
        for (long long i = c0 + threadIdx.x; i < c1; i += 256) atomicAdd(&h[(int)x[i] - vmin], 1u);
    }
    PREP_READS_END
}

// value of the order statistic of rank k (0-based) given the histogram: smallest v with cum(v) > k
// wave-cooperative; returns the same value in every lane
__device__ int order_stat(const uint32_t *h, int vmin, int range, long long k, int lane)
{
    long long before = 0;
    for (int base = 0; base < range; base += 64) {
        const unsigned c = base + lane < range ? h[base + lane] : 0u;
        if (__ballot(c != 0) == 0ull) continue; // empty 64-bin group (most of the int16 range)
        // inclusive scan over the wave
        unsigned long long inc = c;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long up = __shfl_up(inc, o);
            if (lane >= o) inc += up;
        }
        const unsigned long long total = __shfl(inc, 63);
        if (before + (long long)total > k) {
            const unsigned long long hit = __ballot(before + (long long)inc > k);
            return base + __builtin_ctzll(hit) + vmin;
        }
        before += (long long)total;
    }
    return vmin + range - 1;
}

__device__ double np_lerp(double a, double b, double t)
{
    const double diff = b - a;
    double r = a + diff * t;
    if (t >= 0.5) r = b - diff * (1.0 - t);
    return r;
}

// shift and scale of a read from its histogram (normalize_signal_mad, src/schemas/fast5.py:104-114): wave-cooperative;
// shift is valid in every lane, scale in lane 0.  n > 0.
__device__ __forceinline__ void read_stats(const uint32_t *h, int vmin, int vmax, long long n, int lane, double &shift_out, double &scale_out)
{
    const int range = vmax - vmin + 1;
    // np.percentile(x, (46.5, 53.5)), method 'linear'
    double pct[2];
    const double qs[2] = {46.5 / 100.0, 53.5 / 100.0};
    for (int t = 0; t < 2; t++) {
        const double vi = (double)(n - 1) * qs[t];
        long long prev = (long long)floor(vi), next = prev + 1;
        if (vi >= (double)(n - 1)) prev = next = n - 1;
        if (vi < 0) prev = next = 0;
        const double gamma = vi - floor(vi);
        const double xa = (double)order_stat(h, vmin, range, prev, lane), xb = (double)order_stat(h, vmin, range, next, lane);
        pct[t] = np_lerp(xa, xb, gamma);
    }
    const double shift = ((0.0 + pct[0]) + pct[1]) / 2.0;
    // median(|x - shift|): walk outwards from shift over the occupied bins, smaller distance first
    double scale = 0.0;
    if (lane == 0) {
        long long need_hi = n / 2, need_lo = (n % 2) ? n / 2 : n / 2 - 1; // ranks of the middle order statistics
        int lo = (int)floor(shift), hi = lo + 1;                          // lo <= shift < hi
        if (lo > vmax) { lo = vmax; hi = vmax + 1; }
        if (lo < vmin - 1) { lo = vmin - 1; hi = vmin; }
        long long seen = 0;
        double d_lo = 0.0, d_hi = 0.0;
        bool got_lo = false, got_hi = false;
        while (!got_hi && (lo >= vmin || hi <= vmax)) {
            const double dl = lo >= vmin ? fabs((double)lo - shift) : __builtin_huge_val();
            const double dh = hi <= vmax ? fabs((double)hi - shift) : __builtin_huge_val();
            int v;
            double d;
            if (dl <= dh) {
                v = lo--;
                d = dl;
            } else {
                v = hi++;
                d = dh;
            }
            const long long c = h[v - vmin];
            if (c == 0) continue;
            if (!got_lo && seen + c > need_lo) {
                d_lo = d;
                got_lo = true;
            }
            if (seen + c > need_hi) {
                d_hi = d;
                got_hi = true;
            }
            seen += c;
        }
        scale = (n % 2) ? (0.0 + d_hi) / 1.0 : ((0.0 + d_lo) + d_hi) / 2.0;
    }
    shift_out = shift;
    scale_out = scale;
}

__global__ __launch_bounds__(64) void stats_kernel(PrepArgs a)
{
    PREP_READS_BEGIN(a, r)
    const int lane = threadIdx.x;
    const long long n = a.roff[r + 1] - a.roff[r];
    const uint32_t *h = a.hist + (size_t)r * 65536;
    if (n <= 0) {
        if (lane == 0) {
            a.shift_scale[2 * r] = 0.0;
            a.shift_scale[2 * r + 1] = 1.0;
        }
        continue;
    }
    double shift, scale;
    read_stats(h, a.mm[2 * r], a.mm[2 * r + 1], n, lane, shift, scale);
    if (lane == 0) {
        a.shift_scale[2 * r] = shift;
        a.shift_scale[2 * r + 1] = scale;
    }
    PREP_READS_END
}

// Short reads (the STR segments themselves: a few thousand samples): the whole loader in ONE wavefront per read, everything
// in LDS -- working copy, outlier bits, histogram over the occupied value range -- and one pass over HBM (2 bytes in,
// 8 bytes out per sample).  The general kernels above are built for whole reads of 10^4..10^6 samples; on 2000-sample
// segments their per-read blocks spent 4.3 ms per 100k reads on launch granularity and serial walks.  Same arithmetic, same
// results: outlier chains are repaired serially by the lane that finds their head; the histogram becomes its own prefix
// sum, so an order statistic is a search and the median absolute deviation a search over candidate distances -- the k-th
// smallest of the multiset {|v - shift|} does not depend on the order in which equal distances are visited, so the
// two-sided walk of read_stats and this search give the same double.  A read that is longer than `cap` samples or spans more
// than PREP_SHORT_BINS values is left to the general kernels (done[r] = 0).
#define PREP_SHORT_BINS 2048
__global__ __launch_bounds__(64) void short_read_kernel(PrepArgs a, int cap)
{
    extern __shared__ int16_t s_x[];          // cap samples (cap is a multiple of 64)
    uint32_t *s_ol = (uint32_t *)(s_x + cap); // cap / 32 outlier bits (by the RAW values)
    uint32_t *P = s_ol + cap / 32;            // PREP_SHORT_BINS counters, then their inclusive prefix sums
    const int r = blockIdx.x, lane = threadIdx.x;
    const long long len_ll = a.roff[r + 1] - a.roff[r];
    if (len_ll > cap || len_ll <= 0) {
        if (lane == 0) a.done[r] = 0;
        return;
    }
    const int len = (int)len_ll;
    const int16_t *raw = a.raw + a.roff[r];
    const bool brute = a.ol_list != nullptr;
    int lo = 32767, hi = -32768;
    for (int w = lane; w < cap / 32; w += 64) s_ol[w] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    auto take = [&](int i, int v) { // sample i of the read
        s_x[i] = (int16_t)v;
        lo = min(lo, v);
        hi = max(hi, v);
        if (brute && i > 2 && (v > 1000 || v < 250)) atomicOr(&s_ol[i >> 5], 1u << (i & 31)); // rare
    };
    if (a.vec) { // eight samples per 16-byte load, groups aligned in buffer coordinates (as the streaming passes above)
        const long long ro = a.roff[r];
        for_each_sample8(a.raw, ro, ro + len, lane, 64, [&](short8 v, long long base, int first, int last) {
#pragma unroll
            for (int e = 0; e < 8; e++)
                if (e >= first && e < last) take((int)(base + e - ro), (int)v[e]);
        });
    } else {
        for (int i = lane; i < len; i += 64) take(i, (int)raw[i]);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, __shfl_xor(lo, o));
        hi = max(hi, __shfl_xor(hi, o));
    }
    const int vmin = lo, vmax = hi, range = vmax - vmin + 1;
    if (range > PREP_SHORT_BINS) {
        if (lane == 0) a.done[r] = 0;
        return;
    }
    for (int b = lane; b < range; b += 64) P[b] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // spike removal: the outlier that heads a chain (neither of the two samples before it is one) repairs the chain
    auto is_ol = [&](int i) { return i >= 0 && ((s_ol[i >> 5] >> (i & 31)) & 1u); };
    bool any_ol = false;
    for (int w = lane; w < (len + 31) / 32; w += 64) any_ol = any_ol || s_ol[w] != 0u;
    if (brute && __ballot(any_ol)) {
        for (int i = lane; i < len; i += 64) {
            if (!is_ol(i) || is_ol(i - 1) || is_ol(i - 2)) continue;
            int j = i;
            while (true) {
                int w[5];
                int cnt = 0;
                for (int q = j - 2; q < j + 3 && q < len; q++) w[cnt++] = s_x[q];
                s_x[j] = median_small(w, cnt);
                if (j + 1 < len && is_ol(j + 1)) j = j + 1;
                else if (j + 2 < len && is_ol(j + 2)) j = j + 2;
                else break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    for (int i = lane; i < len; i += 64) atomicAdd(&P[(int)s_x[i] - vmin], 1u);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // counters -> inclusive prefix sums, in place
    {
        unsigned int carry = 0u;
        for (int base = 0; base < range; base += 64) {
            unsigned int inc = base + lane < range ? P[base + lane] : 0u;
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned int up = __shfl_up(inc, o);
                if (lane >= o) inc += up;
            }
            inc += carry;
            if (base + lane < range) P[base + lane] = inc;
            carry = __shfl(inc, 63);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    auto cum = [&](int v) -> long long { // samples with value <= v
        return v < vmin ? 0ll : (long long)P[(v > vmax ? vmax : v) - vmin];
    };
    auto order_value = [&](long long k) -> int { // value of the order statistic of rank k (0-based); same in every lane
        for (int base = 0; base < range; base += 64) {
            const unsigned long long hit = __ballot(base + lane < range && (long long)P[base + lane] > k);
            if (hit) return vmin + base + __builtin_ctzll(hit);
        }
        return vmax;
    };
    const long long n = len;
    double pct[2];
    const double qs[2] = {46.5 / 100.0, 53.5 / 100.0};
    for (int t = 0; t < 2; t++) { // np.percentile(x, (46.5, 53.5)), method 'linear' (as read_stats)
        const double vi = (double)(n - 1) * qs[t];
        long long prev = (long long)floor(vi), next = prev + 1;
        if (vi >= (double)(n - 1)) prev = next = n - 1;
        if (vi < 0) prev = next = 0;
        const double gamma = vi - floor(vi);
        pct[t] = np_lerp((double)order_value(prev), (double)order_value(next), gamma);
    }
    const double shift = ((0.0 + pct[0]) + pct[1]) / 2.0;
    // median(|x - shift|): the distance of rank k is the smallest candidate |v - shift| with more than k samples within it.
    // Lanes try the candidates of one side, 64 at a time, nearest first; "within rho" is an interval of values whose ends
    // are found arithmetically and corrected with the very comparison fabs((double)v - shift) <= rho.
    int fl = (int)floor(shift); // fl <= shift < fl + 1
    if (fl > vmax) fl = vmax;
    if (fl < vmin - 1) fl = vmin - 1;
    auto within = [&](double rho) -> long long { // samples with |v - shift| <= rho
        int a0 = (int)ceil(shift - rho), b0 = (int)floor(shift + rho);
        a0 = a0 < vmin - 1 ? vmin - 1 : (a0 > vmax + 1 ? vmax + 1 : a0);
        b0 = b0 < vmin - 1 ? vmin - 1 : (b0 > vmax + 1 ? vmax + 1 : b0);
        while (a0 - 1 >= vmin && fabs((double)(a0 - 1) - shift) <= rho) a0--;
        while (a0 <= vmax && !(fabs((double)a0 - shift) <= rho)) a0++;
        while (b0 + 1 <= vmax && fabs((double)(b0 + 1) - shift) <= rho) b0++;
        while (b0 >= vmin && !(fabs((double)b0 - shift) <= rho)) b0--;
        return b0 >= a0 ? cum(b0) - cum(a0 - 1) : 0ll;
    };
    auto dist_of_rank = [&](long long k) -> double {
        double best = __builtin_huge_val();
        for (int side = 0; side < 2; side++) {
            const int first = side == 0 ? fl : fl + 1, count = side == 0 ? fl - vmin + 1 : vmax - fl; // candidates first, first -+ 1, ..
            for (int j0 = 0; j0 < count; j0 += 64) {
                const int j = j0 + lane;
                const int v = side == 0 ? first - j : first + j;
                const double rho = fabs((double)v - shift);
                const bool ok = j < count && within(rho) > k;
                const unsigned long long hit = __ballot(ok);
                if (hit) {
                    const int jj = j0 + __builtin_ctzll(hit);
                    const int vv = side == 0 ? first - jj : first + jj;
                    const double d = fabs((double)vv - shift);
                    best = d < best ? d : best;
                    break;
                }
            }
        }
        return best;
    };
    const double d_hi = dist_of_rank(n / 2), d_lo = (n % 2) ? d_hi : dist_of_rank(n / 2 - 1);
    const double scale = (n % 2) ? (0.0 + d_hi) / 1.0 : ((0.0 + d_lo) + d_hi) / 2.0;
    if (lane == 0) {
        a.shift_scale[2 * r] = shift;
        a.shift_scale[2 * r + 1] = scale;
        a.done[r] = 1;
    }
    long long slo = a.seg_lo[r], shi = a.seg_hi[r] + 1; // python slice [lo:hi)
    if (slo < 0) slo += len;
    if (slo < 0) slo = 0;
    if (slo > len) slo = len;
    if (shi < 0) shi += len;
    if (shi < 0) shi = 0;
    if (shi > len) shi = len;
    const int cnt = (int)(a.ooff[r + 1] - a.ooff[r]);
    double *out = a.out + a.ooff[r];
    // (x - shift) / scale with the part of the compiler's fp64 division that depends on the denominator alone done once per
    // read (mid_kernels.hip: wsx_div_by -- the same three last operations, bit-identical while v_div_scale would not rescale:
    // |x - shift| is 0 or in [0.25, 2^17], scale in [0.25, 2^17]; a zero or non-finite scale takes the plain division)
    if (scale >= 0.25 && scale <= 131072.0) {
        const double r0 = __builtin_amdgcn_rcp(scale);
        const double f0 = __builtin_fma(-scale, r0, 1.0);
        const double r1 = __builtin_fma(r0, f0, r0);
        const double f1 = __builtin_fma(-scale, r1, 1.0);
        const double rr = __builtin_fma(r1, f1, r1);
        for (int i = lane; i < cnt; i += 64) {
            const double x = (double)s_x[slo + i] - shift;
            const double q = x * rr;
            const double e = __builtin_fma(-scale, q, x);
            out[i] = __builtin_fma(e, rr, q);
        }
    } else {
        for (int i = lane; i < cnt; i += 64) out[i] = ((double)s_x[slo + i] - shift) / scale;
    }
}

__global__ __launch_bounds__(256) void norm_kernel(PrepArgs a)
{
    PREP_READS_BEGIN(a, r)
    const long long len = a.roff[r + 1] - a.roff[r];
    long long lo = a.seg_lo[r], hi = a.seg_hi[r] + 1; // python slice [lo:hi)
    if (lo < 0) lo += len;
    if (lo < 0) lo = 0;
    if (lo > len) lo = len;
    if (hi < 0) hi += len;
    if (hi < 0) hi = 0;
    if (hi > len) hi = len;
    const long long cnt = a.ooff[r + 1] - a.ooff[r]; // == max(hi - lo, 0), computed by the host the same way
    const double shift = a.shift_scale[2 * r], scale = a.shift_scale[2 * r + 1];
    const int16_t *x = a.clean + a.roff[r] + lo;
    double *out = a.out + a.ooff[r];
    for (long long i = (long long)blockIdx.y * 256 + threadIdx.x; i < cnt; i += (long long)gridDim.y * 256)
        out[i] = ((double)x[i] - shift) / scale;
    PREP_READS_END
}

} // namespace

int wsx_internal_device(wsx_caller *c);
hipStream_t wsx_internal_stream(wsx_caller *c);
void wsx_internal_set_error(const char *msg);
extern "C" int wsx_internal_on_exception(void);
hipError_t wsx_internal_prep_buffer(wsx_caller *c, int slot, size_t bytes, void **p);
uint64_t wsx_internal_workspace_limit(wsx_caller *c);
// pinned host memory of the handle for this function's metadata (>= bytes), and the event recorded after its last use
hipError_t wsx_internal_prep_pinned(wsx_caller *c, size_t bytes, void **p, hipEvent_t *last_use);

#define PCHK(expr)                                                                                       \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) {                                                                          \
            char b_[512];                                                                                \
            snprintf(b_, sizeof(b_), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            wsx_internal_set_error(b_);                                                                  \
            return WSX_ERR_HIP;                                                                          \
        }                                                                                                \
    } while (0)

extern "C" int wsx_prepare_signals(wsx_caller *c, int mem, const int16_t *raw, const int64_t *raw_offsets,
                                   const int64_t *seg_start, const int64_t *seg_end, int64_t n_reads, int32_t spike_removal,
                                   double *signal_out, const int64_t *out_offsets, double *shift_scale)
try {
    if (!c || !raw_offsets || !seg_start || !seg_end || !out_offsets || n_reads < 0 || (n_reads > 0 && (!raw || !signal_out))) {
        wsx_internal_set_error("wsx_prepare_signals: null argument");
        return WSX_ERR_INVALID;
    }
    if (spike_removal < 0 || spike_removal > 3) {
        wsx_internal_set_error("spike_removal: 0 = None, 1 = Brute, 2 = median3, 3 = median5 (src/schemas/fast5.py:68-75)");
        return WSX_ERR_INVALID;
    }
    if (mem != WSX_MEM_HOST && mem != WSX_MEM_DEVICE) {
        wsx_internal_set_error("mem must be WSX_MEM_HOST or WSX_MEM_DEVICE");
        return WSX_ERR_INVALID;
    }
    for (int64_t r = 0; r < n_reads; r++) {
        const int64_t len = raw_offsets[r + 1] - raw_offsets[r];
        int64_t lo = seg_start[r], hi = seg_end[r] + 1;
        if (len < 0) {
            wsx_internal_set_error("raw_offsets must be non-decreasing");
            return WSX_ERR_INVALID;
        }
        if (lo < 0) lo += len;
        lo = std::min(std::max<int64_t>(lo, 0), len);
        if (hi < 0) hi += len;
        hi = std::min(std::max<int64_t>(hi, 0), len);
        if (out_offsets[r + 1] - out_offsets[r] != std::max<int64_t>(hi - lo, 0)) {
            wsx_internal_set_error("out_offsets do not match the slices [l_start_raw : r_end_raw + 1]");
            return WSX_ERR_INVALID;
        }
    }
    if (n_reads == 0) return WSX_SUCCESS;
    PCHK(hipSetDevice(wsx_internal_device(c)));
    hipStream_t st = wsx_internal_stream(c);
    const bool host = mem == WSX_MEM_HOST;
    // a histogram of 65 536 counters (256 KiB) per read, of which only the read's occupied value range is ever touched: half
    // the handle's workspace limit goes to them (default 16 GiB -> 32 768 reads per chunk; 288 GB of HBM take 100k short
    // reads in one go, where 4 096-read chunks spent most of a call on launches and synchronisation)
    // (at most 32 GiB of them: 131 072 reads per chunk)
    const int64_t chunk_reads = std::max<int64_t>(4096, (int64_t)(std::min<uint64_t>(wsx_internal_workspace_limit(c) / 2, 32ull << 30) / (65536 * 4)));
    // Metadata (chunk-relative offsets, slice bounds, initial value ranges) goes through pinned memory owned by the handle, so
    // that the uploads are asynchronous and nothing of the caller's is referenced after the return; device-buffer calls then
    // return as soon as the work is enqueued on the handle's stream.
    const int64_t n_chunks = (n_reads + chunk_reads - 1) / chunk_reads;
    char *pin = nullptr;
    hipEvent_t pin_event = nullptr;
    PCHK(wsx_internal_prep_pinned(c, (size_t)(5 * n_reads + 2 * n_chunks) * 8 + 64, (void **)&pin, &pin_event));
    PCHK(hipEventSynchronize(pin_event)); // the previous call's uploads have left the buffer (a no-op before the first call)
    size_t pin_used = 0;
    uint32_t *d_hist = nullptr;
    // every device buffer of this function lives in the handle's pool (slot 0: histograms, 1..: per-chunk buffers): a
    // fresh hipMalloc / hipFree of ~1.5 GB per call cost a third of the call
    PCHK(wsx_internal_prep_buffer(c, 0, (size_t)std::min<int64_t>(n_reads, chunk_reads) * 65536 * 4, (void **)&d_hist));
    for (int64_t f = 0; f < n_reads; f += chunk_reads) {
        const int64_t cnt = std::min(chunk_reads, n_reads - f);
        const int64_t rbase = raw_offsets[f], rsz = raw_offsets[f + cnt] - rbase;
        const int64_t obase = out_offsets[f], osz = out_offsets[f + cnt] - obase;
        int64_t *h_roff = (int64_t *)(pin + pin_used), *h_ooff = h_roff + cnt + 1, *h_lo = h_ooff + cnt + 1, *h_hi = h_lo + cnt;
        int32_t *h_mm = (int32_t *)(h_hi + cnt);
        pin_used += (size_t)(5 * cnt + 2) * 8;
        int64_t max_len = 0;
        for (int64_t r = 0; r <= cnt; r++) {
            h_roff[r] = raw_offsets[f + r] - rbase;
            h_ooff[r] = out_offsets[f + r] - obase;
            if (r < cnt) max_len = std::max(max_len, raw_offsets[f + r + 1] - raw_offsets[f + r]);
        }
        int16_t *d_raw = nullptr, *d_clean = nullptr;
        int64_t *d_meta = nullptr;
        double *d_ss = nullptr, *d_out = nullptr;
        int next_slot = 1;
        auto alloc = [&](void **p, size_t bytes) -> hipError_t {
            return wsx_internal_prep_buffer(c, next_slot++, std::max<size_t>(bytes, 256), p);
        };
        if (host) {
            PCHK(alloc((void **)&d_raw, (size_t)rsz * 2 + 16));
            PCHK(hipMemcpyAsync(d_raw, raw + rbase, (size_t)rsz * 2, hipMemcpyHostToDevice, st));
            PCHK(alloc((void **)&d_out, (size_t)osz * 8));
        } else {
            d_raw = const_cast<int16_t *>(raw) + rbase;
            d_out = signal_out + obase;
        }
        PCHK(alloc((void **)&d_clean, (size_t)rsz * 2 + 16));
        PCHK(alloc((void **)&d_meta, (size_t)(4 * cnt + 2) * 8));
        PCHK(alloc((void **)&d_ss, (size_t)cnt * 16));
        int64_t *d_roff = d_meta, *d_ooff = d_meta + cnt + 1, *d_lo = d_ooff + cnt + 1, *d_hi = d_lo + cnt;
        memcpy(h_lo, seg_start + f, (size_t)cnt * 8);
        memcpy(h_hi, seg_end + f, (size_t)cnt * 8);
        for (int64_t r = 0; r < cnt; r++) {
            h_mm[2 * r] = 32767; // empty reads keep an empty (negative) range
            h_mm[2 * r + 1] = -32768;
        }
        int32_t *d_mm = nullptr;
        PCHK(alloc((void **)&d_mm, (size_t)cnt * 8));
        // (offsets, offsets, bounds, bounds are contiguous in the pinned buffer and on the device: one copy)
        PCHK(hipMemcpyAsync(d_roff, h_roff, (size_t)(4 * cnt + 2) * 8, hipMemcpyHostToDevice, st));
        PCHK(hipMemcpyAsync(d_mm, h_mm, (size_t)cnt * 8, hipMemcpyHostToDevice, st));
        // eight-sample loads need 16-byte aligned buffers (groups of eight are aligned in buffer coordinates, reads may start
        // anywhere) and must not run past the allocation: the staging buffers are padded, a caller's device buffer is only
        // read in whole groups when it is aligned and the last group ends inside the chunk
        const int vec = ((uintptr_t)d_raw % 16 == 0 && (uintptr_t)d_clean % 16 == 0 && (host || rsz % 8 == 0)) ? 1 : 0;
        // outlier lists for the spike pass: per read room for one sample in 16 (real reads: a fraction of a percent)
        long long *d_ol = nullptr;
        unsigned int *d_olc = nullptr;
        if (spike_removal == 1) {
            const size_t entries = (size_t)(rsz / 16 + 64 * cnt + 64);
            PCHK(alloc((void **)&d_ol, entries * 8 + (size_t)cnt * 4 + 16));
            d_olc = (unsigned int *)(d_ol + entries);
            PCHK(hipMemsetAsync(d_olc, 0, (size_t)cnt * 4, st));
        }
        uint8_t *d_done = nullptr;
        PCHK(alloc((void **)&d_done, (size_t)cnt));
        PrepArgs a{d_raw, d_clean, d_roff, d_lo, d_hi, d_ooff, d_hist, d_mm, d_ss, d_out, (int)cnt, (int)max_len, vec, d_ol, d_olc, d_done, 1};
        int64_t max_out = 0;
        for (int64_t r = 0; r < cnt; r++) max_out = std::max(max_out, h_ooff[r + 1] - h_ooff[r]);
        // (a chunk of short reads: the general kernels mostly find done[r] set -- one block per read is enough for the rest)
        const unsigned gn = max_len <= 8192 ? 1u : (unsigned)std::min<int64_t>(std::max<int64_t>((max_out + 255) / 256, 1), 4096);
        const unsigned gc = (unsigned)std::max<int64_t>((max_len + PREP_CHUNK - 1) / PREP_CHUNK, 1);
        const unsigned gs = (unsigned)std::max<int64_t>((max_len + PREP_STREAM - 1) / PREP_STREAM, 1);
        {
            // short reads first, one block each (the kernel decides per read and tells the general kernels what is left)
            int64_t min_len = max_len;
            for (int64_t r = 0; r < cnt; r++) min_len = std::min(min_len, h_roff[r + 1] - h_roff[r]);
            const int cap = (int)std::min<int64_t>(((std::min<int64_t>(max_len, 8192) + 63) / 64) * 64, 8192);
            const size_t lds = (size_t)cap * 2 + (size_t)cap / 8 + 8 + PREP_SHORT_BINS * 4;
            if (min_len <= cap && spike_removal <= 1 && !wsx_exp_env("WSX_PREP_GENERAL")) {
                hipLaunchKernelGGL(short_read_kernel, dim3((unsigned)cnt), dim3(64), lds, st, a, cap);
                // every read could go that way: what is left for the general kernels is the odd read with a wide value range
                if (max_len <= cap && !wsx_exp_env("WSX_PREP_NO_GROUPS")) a.group = 256;
            } else {
                PCHK(hipMemsetAsync(d_done, 0, (size_t)cnt, st)); // no read of this chunk is short
            }
        }
        const unsigned gx = (unsigned)((cnt + a.group - 1) / a.group); // blocks along x: one per read, or per group of reads
        if (spike_removal == 2) hipLaunchKernelGGL(medfilt_kernel<3>, dim3(gx, gs), dim3(256), 0, st, a);
        else if (spike_removal == 3) hipLaunchKernelGGL(medfilt_kernel<5>, dim3(gx, gs), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(copy_kernel, dim3(gx, gs), dim3(256), 0, st, a);
        hipLaunchKernelGGL(zero_kernel, dim3(gx, max_len <= 8192 ? 1 : 8), dim3(256), 0, st, a);
        if (spike_removal == 1) {
            if (d_ol) hipLaunchKernelGGL(spike_list_kernel, dim3(gx), dim3(256), 0, st, a);
            hipLaunchKernelGGL(spike_kernel, dim3(gx, gs), dim3(256), 0, st, a); // returns at once unless the list overflowed
        }
        hipLaunchKernelGGL(hist_kernel, dim3(gx, gc), dim3(256), a.group > 1 ? 0 : PREP_LDS_BINS * sizeof(uint32_t), st, a);
        hipLaunchKernelGGL(stats_kernel, dim3(gx), dim3(64), 0, st, a);
        hipLaunchKernelGGL(norm_kernel, dim3(gx, gn), dim3(256), 0, st, a);
        PCHK(hipGetLastError());
        if (host) PCHK(hipMemcpyAsync(signal_out + obase, d_out, (size_t)osz * 8, hipMemcpyDeviceToHost, st));
        if (shift_scale) {
            PCHK(hipMemcpyAsync(shift_scale + 2 * f, d_ss, (size_t)cnt * 16, host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, st));
        }
        if (host) PCHK(hipStreamSynchronize(st)); // the caller's host buffers are complete when the call returns
    }
    PCHK(hipEventRecord(pin_event, st));
    return WSX_SUCCESS;
} catch (...) {
    return wsx_internal_on_exception();
}
