// mid_kernels.hip -- everything between the two DTW passes, and the per-read epilogue.
//
// Upstream functions restated here for the device (paths relative to the upstream repository):
//   WarpResult.create_alignment / StateAlignment  src/caller/caller.py:17-43,65-96   (run statistics)
//   filter_alignment / rescale_signal            src/caller/caller.py:304-318       (sort + FITPACK cubic)
//   mask_bad_repeats / find_event_borders / segment / calc_ttest / check_segments / mask_big_events
//                                                src/caller/caller.py:330-421
//   WarpSTR._get_sequence (length only)          src/caller/caller.py:178-187
//   state-wise cost                              src/caller/caller.py:138-139
// Arithmetic that upstream delegates to NumPy (pairwise-summed mean/std) and SciPy FITPACK
// (curfit with s = m: least-squares cubic by Givens rotations; splev) is reproduced operation by
// operation in fp64 (compiled with -ffp-contract=off), see oracle/warpstr_oracle.c for the CPU twin
// used by the parity tests.
//
// Parallelisation: flat kernels, no serialised phases (see the kernel list further down):
//   run_stats_kernel   thread per run             borders_kernel   wavefront per read
//   tstat_kernel       thread per sample          chunk_kernel     thread per (read, chunk)
//   sort_kernel        wavefront per read         fit_kernel       quad of lanes per read (pipelined Givens recurrence)
//   eval_kernel        thread per sample (de Boor evaluation, 6 divisions)
#include <algorithm>
#include <cstdlib>

#include "../../include/warpstr_hip.h"
#include "wsx_device.h"

namespace {

constexpr double kInf = __builtin_huge_val();

// ---- NumPy pairwise summation (n < 8: plain loop; n <= 128: 8 accumulators; else split) --------
struct LoadPlain {
    const double *a;
    __device__ double operator()(int i) const { return a[i]; }
};
struct LoadSqDev {
    const double *a;
    double mean;
    __device__ double operator()(int i) const
    {
        const double d = a[i] - mean;
        return d * d;
    }
};
template <class L>
__device__ __forceinline__ double pw_block_inl(const L &ld, int o, int n) // n <= 128
{
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; i++) res += ld(o + i);
        return res;
    }
    double r0 = ld(o), r1 = ld(o + 1), r2 = ld(o + 2), r3 = ld(o + 3), r4 = ld(o + 4), r5 = ld(o + 5), r6 = ld(o + 6),
           r7 = ld(o + 7);
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
        r0 += ld(o + i);
        r1 += ld(o + i + 1);
        r2 += ld(o + i + 2);
        r3 += ld(o + i + 3);
        r4 += ld(o + i + 4);
        r5 += ld(o + i + 5);
        r6 += ld(o + i + 6);
        r7 += ld(o + i + 7);
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; i++) res += ld(o + i);
    return res;
}

// The recursion  sum(a,n) = sum(a,n2) + sum(a+n2,n-n2),  n2 = n/2 - (n/2)%8,  of NumPy's pairwise summation for n > 128,
// walked leaf by leaf without a call and without a local array: `path` says, per depth, whether the walk is in a right
// child; the finished left siblings wait in five registers (enough for 2048 elements: each half is at most n/2 + 7 long)
// and, deeper down, in GLOBAL memory the caller points at (>= 27 doubles owned by the calling thread; only inputs of more
// than 2048 elements -- a run of that many samples, that many runs in a repeat -- ever touch it).  A local stack array or a
// non-inlined helper would give the whole kernel a private segment, and the runtime reserves that segment for every
// wave slot of the device on every stream: 448 bytes made 0.9 GB of device memory disappear on the first call.
template <class L>
__device__ __forceinline__ double pw_tree(const L &ld, int n, double *gstack)
{
    double l1 = 0, l2 = 0, l3 = 0, l4 = 0, l5 = 0;
    auto put = [&](int d, double v) {
        if (d == 1) l1 = v;
        else if (d == 2) l2 = v;
        else if (d == 3) l3 = v;
        else if (d == 4) l4 = v;
        else if (d == 5) l5 = v;
        else gstack[d] = v;
    };
    auto get = [&](int d) -> double { return d == 1 ? l1 : d == 2 ? l2 : d == 3 ? l3 : d == 4 ? l4 : d == 5 ? l5 : gstack[d]; };
    unsigned int path = 0u;
    int d = 0, o = 0, nn = n;
    while (true) {
        while (nn > 128) { // down to the leftmost leaf below this node
            int n2 = nn / 2;
            n2 -= n2 % 8;
            d++;
            path &= ~(1u << d);
            nn = n2;
        }
        double ret = pw_block_inl(ld, o, nn);
        while (d > 0 && ((path >> d) & 1u)) { // a right child: left sibling + this, and up
            ret = get(d) + ret;
            d--;
        }
        if (d == 0) return ret;
        put(d, ret); // a left child: wait for the right sibling
        path |= 1u << d;
        o = 0;
        nn = n;
        for (int e = 1; e <= d; e++) { // offset and length of that sibling: from the root along the path
            int n2 = nn / 2;
            n2 -= n2 % 8;
            if ((path >> e) & 1u) {
                o += n2;
                nn -= n2;
            } else {
                nn = n2;
            }
        }
    }
}

template <class L>
__device__ __forceinline__ double np_pairwise_sum(const L &ld, int n, double *gstack)
{
    if (n <= 128) return pw_block_inl(ld, 0, n);
    return pw_tree(ld, n, gstack);
}

// mean and std of one run (np.average / np.std of the run's samples)
// gstack: >= 27 doubles of global scratch for this thread, touched only for runs of more than 2048 samples
__device__ __forceinline__ void run_mean_std(const double *sig, int s0, int len, double &mean, double &sd, double *gstack)
{
    mean = np_pairwise_sum(LoadPlain{sig + s0}, len, gstack) / (double)len;
    sd = sqrt(np_pairwise_sum(LoadSqDev{sig + s0, mean}, len, gstack) / (double)len);
}

// k-th order statistic (0-based) of a[0..n) by rank counting; ties broken by index.
__device__ __forceinline__ double select_rank(const double *a, int n, int kth)
{
    for (int i = 0; i < n; i++) {
        const double x = a[i];
        int rank = 0;
        for (int q = 0; q < n; q++) {
            const double y = a[q];
            rank += (y < x) || (y == x && q < i);
        }
        if (rank == kth) return x;
    }
    return a[0];
}

__device__ __forceinline__ double np_median(const double *a, int n)
{
    if (n & 1) return select_rank(a, n, n / 2);
    return (select_rank(a, n, n / 2 - 1) + select_rank(a, n, n / 2)) / 2.0;
}

// ---- fp64 division by a denominator that many divisions share ---------------------------------
// The compiler expands x / d into v_div_scale (x2), v_rcp_f64, four FMAs that refine the reciprocal, a multiply, a
// residual FMA, v_div_fmas and v_div_fixup.  Everything up to the refined reciprocal depends on d alone whenever
// v_div_scale leaves its operands unscaled, i.e. for d and x far from the ends of the exponent range (and x = 0) --
// which holds for every use below: d is 3.0 or the span of a read's run means, x a sum of O(1) signal terms or a
// B-spline weight of at least 2^-160 in magnitude.  wsx_div_by() then performs the remaining three operations of that
// very expansion, so the quotient is bit-identical to x / d (checked bit for bit against the plain form on 100k reads:
// scripts/exp_bitcmp.py) at a quarter of the instructions.
__device__ __forceinline__ double wsx_recip_refined(double d)
{
    const double r = __builtin_amdgcn_rcp(d);
    const double f0 = __builtin_fma(-d, r, 1.0);
    const double f1 = __builtin_fma(r, f0, r);
    const double f2 = __builtin_fma(-d, f1, 1.0);
    return __builtin_fma(f1, f2, f1);
}
__device__ __forceinline__ double wsx_div_by(double x, double d, double r)
{
    const double q = x * r;
    const double e = __builtin_fma(-d, q, x);
    return __builtin_fma(e, r, q);
}

// ---- sliding t-test segmentation (caller.py:347-378) ------------------------------------------
__device__ __forceinline__ double mean3v(double a0, double a1, double a2, double r3)
{
    return wsx_div_by(((0.0 + a0) + a1) + a2, 3.0, r3);
}
__device__ __forceinline__ double std3v(double a0, double a1, double a2, double mu, double r3)
{
    const double d0 = a0 - mu, d1 = a1 - mu, d2 = a2 - mu;
    return sqrt(wsx_div_by(((0.0 + d0 * d0) + d1 * d1) + d2 * d2, 3.0, r3));
}

// Python slice bounds [a:b] on a sequence of length n
__device__ void py_slice(long long a, long long b, long long n, int *lo, int *hi)
{
    if (a < 0) a += n;
    if (a < 0) a = 0;
    if (a > n) a = n;
    if (b < 0) b += n;
    if (b < 0) b = 0;
    if (b > n) b = n;
    *lo = (int)a;
    *hi = (int)(b < a ? a : b);
}

// ------------------------------------------------------------------------------------------------
// The stage between the DTW passes as FLAT kernels (each massively parallel, no serialised phases):
//   run_stats_kernel   one thread per run            : alignment records (mean/std/good/cost)
//   borders_kernel     one wavefront per read        : find_event_borders, status, cost, allele length
//   tstat_kernel       one thread per sample         : sliding-window statistics and t-statistic; zeroes the mask
//   chunk_kernel       one thread per (read, chunk)  : segment() peak scan -> bad-repeat mask
//   sort_kernel        one wavefront per read        : stable sort of the accepted (value, expected) pairs
// ------------------------------------------------------------------------------------------------
struct ReadView {
    int r, lr, T, n;
    long long off;
    const uint16_t *rs;
    const int32_t *rst;
    __device__ int fstate(int k) const { return rs[n - 1 - k]; }
    __device__ int fstart(int k) const { return rst[n - 1 - k]; }
    __device__ int fend(int k) const { return (k == n - 1) ? T : rst[n - 2 - k]; } // exclusive
};

__device__ __forceinline__ ReadView view(const MidArgs &a, int lr)
{
    ReadView v;
    v.lr = lr;
    v.r = a.first_read + lr;
    v.off = a.offsets[v.r] - a.base_off;
    v.T = (int)(a.offsets[v.r + 1] - a.offsets[v.r]);
    v.n = a.n_runs[lr];
    v.rs = a.run_state + v.off;
    v.rst = a.run_start + v.off;
    return v;
}

// (1) alignment records: one per run (create_alignment, reps_as_one = False), caller.py:17-43,65-96.
// A block handles 64 consecutive runs, i.e. one contiguous span of the signal: the span is loaded coalesced into LDS
// (when it fits) and every thread then walks its own run there.
#ifndef RS_CAP
#define RS_CAP 1024
#endif
#ifndef RS_GROUPS
#define RS_GROUPS 4
#endif //// grid.y: a read's 64-run groups are dealt round-robin to this many single-wave blocks
__global__ __launch_bounds__(64) void run_stats_kernel(MidArgs a)
{
    __shared__ double buf[RS_CAP];
    const int lr = blockIdx.x;
    if (a.status[lr] != 0) return;
    const ReadView v = view(a, lr);
    const int tid = threadIdx.x;
    const DevAutomaton &A = a.aut[a.aut_id[v.r]];
    const double *sig = a.signal + v.off;
    // (a grid with one block per possible group, T/(m-1)/64 of them, is mostly empty blocks: runs last ~9 samples)
    for (int k0 = blockIdx.y * 64; k0 < v.n; k0 += RS_GROUPS * 64) {
        const int klast = (k0 + 64 < v.n ? k0 + 64 : v.n) - 1;
        // Every thread fetches its own run first (start, end, state: three independent loads) and the group's span comes
        // from the first and the last lane -- the staging loads and the expected levels then go out together, so a group
        // costs three dependent memory round trips instead of five.
        const int k = k0 + tid;
        const bool mine = k <= klast;
        const int s0 = mine ? v.fstart(k) : 0, s1 = mine ? v.fend(k) : 0, st = mine ? v.fstate(k) : 0;
        const int s_lo = __builtin_amdgcn_readfirstlane(s0), s_hi = __builtin_amdgcn_readlane(s1, klast - k0);
        const double expd = mine ? A.value[st] : 0.0;
        const bool staged = (s_hi - s_lo) <= RS_CAP;
        if (staged) {
            for (int q = tid; q < s_hi - s_lo; q += 64) buf[q] = sig[s_lo + q];
            __syncthreads();
        }
        if (mine) {
            const int len = s1 - s0;
            double val, sd;
            if (staged && len <= 128) { // common case: straight-line LDS reads
                const double *p = buf + (s0 - s_lo);
                val = pw_block_inl(LoadPlain{p}, 0, len) / (double)len;
                sd = sqrt(pw_block_inl(LoadSqDev{p, val}, 0, len) / (double)len);
            } else {
                // (a run of more than 2048 samples owns that many doubles of the per-sample scratch: room for the stack)
                run_mean_std(sig, s0, len, val, sd, a.scr0 + v.off + s0);
            }
            if (a.prm.method_median) val = np_median(sig + s0, len);
            const bool good = (len >= a.prm.m) && (sd < a.prm.max_std) && (fabs(expd - val) <= a.prm.threshold);
            a.al_value[v.off + k] = val;
            a.al_expected[v.off + k] = expd;
            a.al_cost[v.off + k] = fabs(val - expd);
            a.al_good[v.off + k] = good ? 1 : 0;
        }
        __syncthreads(); // the staging buffer is reused by the next group
    }
}

// (1') rescaling.reps_as_one = True (caller.py:69-79): one record per distinct state on the path, in ascending
// state order, over ALL samples of that state (time order).  Rarely used option: one thread per read, serial.
__global__ __launch_bounds__(64) void reps_stats_kernel(MidArgs a)
{
    const int lr = blockIdx.x * blockDim.x + threadIdx.x;
    if (lr >= a.n_reads) return;
    if (a.status[lr] != 0) return;
    const ReadView v = view(a, lr);
    const DevAutomaton &A = a.aut[a.aut_id[v.r]];
    const int S = A.n_states, n = v.n;
    int32_t *cnt = a.state_scratch + (size_t)lr * 2 * a.max_states, *stoff = cnt + a.max_states;
    const double *sig = a.signal + v.off;
    double *gs = a.scr0 + v.off;
    for (int j = 0; j < S; j++) cnt[j] = 0;
    for (int k = 0; k < n; k++) cnt[v.fstate(k)] += v.fend(k) - v.fstart(k);
    int o = 0;
    for (int j = 0; j < S; j++) {
        stoff[j] = o;
        o += cnt[j];
        cnt[j] = 0;
    }
    for (int k = 0; k < n; k++) {
        const int j = v.fstate(k), s0 = v.fstart(k), len = v.fend(k) - s0;
        const int b = stoff[j] + cnt[j];
        for (int t = 0; t < len; t++) gs[b + t] = sig[s0 + t];
        cnt[j] += len;
    }
    int u = 0;
    for (int j = 0; j < S; j++) {
        const int len = cnt[j];
        if (len == 0) continue;
        double val, sd;
        run_mean_std(gs, stoff[j], len, val, sd, a.scr1 + v.off + stoff[j]);
        if (a.prm.method_median) val = np_median(gs + stoff[j], len);
        const double expd = A.value[j];
        const bool good = (len >= a.prm.m) && (sd < a.prm.max_std) && (fabs(expd - val) <= a.prm.threshold);
        a.al_value[v.off + u] = val;
        a.al_expected[v.off + u] = expd;
        a.al_cost[v.off + u] = fabs(val - expd);
        a.al_good[v.off + u] = good ? 1 : 0;
        u++;
    }
    a.n_align[lr] = u;
}

__device__ __forceinline__ double readlane_f64(double x, int l)
{
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// NumPy's pairwise sum of a[0..n) (n <= 128) with the eight partial sums on lanes 0..7: the same additions in the same
// order as pw_block, the strided accumulations side by side.  Every lane returns the result.
__device__ __forceinline__ double pw_block_wave(const double *a, int n, int lane)
{
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    }
    double r = lane < 8 ? a[lane] : 0.0;
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
        if (lane < 8) r += a[i + lane];
    const double r0 = readlane_f64(r, 0), r1 = readlane_f64(r, 1), r2 = readlane_f64(r, 2), r3 = readlane_f64(r, 3),
                 r4 = readlane_f64(r, 4), r5 = readlane_f64(r, 5), r6 = readlane_f64(r, 6), r7 = readlane_f64(r, 7);
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; i++) res += a[i];
    return res;
}

// (2,3,5) per read: allele length, find_event_borders, chunk range checks, state-wise cost, results.  One wavefront per
// read: the scans over the run list (good records, first/last repeat state, the run that closes the last chunk, the
// chunk range checks) are lane-parallel with wave reductions, the state-wise cost keeps NumPy's summation order.
__global__ __launch_bounds__(256) void borders_kernel(MidArgs a)
{
    const int lane = threadIdx.x & 63;
    const int lr = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (lr >= a.n_reads) return;
    wsx_result *res = (wsx_result *)a.results + lr;
    MidRec rec{};
    int status = a.status[lr];
    if (status != 0) {
        if (lane == 0) {
            res->status = status;
            if (a.pass == 1) {
                res->len1 = res->len2 = res->n_trans1 = res->n_trans2 = 0;
                res->reserved = 0;
                res->cost1 = res->cost2 = __builtin_nan("");
                res->dtw_end_cost1 = res->dtw_end_cost2 = kInf;
            }
            a.rec[lr] = rec;
        }
        return;
    }
    const ReadView v = view(a, lr);
    const DevAutomaton &A = a.aut[a.aut_id[v.r]];
    const int n = v.n, T = v.T, sis = a.prm.states_in_segment;
    const uint8_t *alg = a.al_good + v.off;
    const double *alc = a.al_cost + v.off;

    int slo, shi;
    py_slice((long long)A.flank_length - A.seq_idx[v.fstate(0)], -(long long)A.flank_length, n, &slo, &shi);
    const int seqlen = shi - slo;

    const int na = a.n_align ? a.n_align[lr] : n; // alignment records: one per run, or per distinct state
    // 64 runs per step, one per lane; ballots turn the flags into wave masks: counts and first/last positions are scalar
    int n_good = 0, start = -1, end = -1;
    for (int k0 = 0; k0 < na; k0 += 64) n_good += __builtin_popcountll(__ballot(k0 + lane < na && alg[k0 + lane]));
    for (int k0 = 0; k0 < n; k0 += 64) {
        const unsigned long long rep = __ballot(k0 + lane < n && A.repeat_mask[v.fstate(k0 + lane)]);
        if (rep) {
            if (start < 0) start = k0 + __builtin_ctzll(rep);
            end = k0 + 63 - __builtin_clzll(rep);
        }
    }
    int nsel = 0, nb = 0;
    if (n_good < 4) {
        status = WSX_READ_FIT_POINTS; // rescale_signal runs (and fails) before mask_bad_repeats upstream
    } else if (start < 0) {
        status = WSX_READ_NO_REPEAT;
    } else {
        nb = end - start;
        const int add = (((nb - 1) % sis) + sis) % sis;
        if (add > 0) {
            end = end + (sis - add);
            if (end >= n) {
                status = WSX_READ_SEGMENT_RANGE;
            } else {
                const int es = v.fstate(end);
                int eR = end; // the last run in the same state after `end`, or `end` itself
                for (int k0 = end + 1; k0 < n; k0 += 64) {
                    const unsigned long long same = __ballot(k0 + lane < n && v.fstate(k0 + lane) == es);
                    if (same) eR = k0 + 63 - __builtin_clzll(same);
                }
                nb = eR - start;
            }
        }
        if (status == 0) {
            nsel = nb > 0 ? (nb + sis - 1) / sis : 0;
            if (nsel == 0) status = WSX_READ_SEGMENT_RANGE;
        }
    }
    if (status == 0) {
        for (int c0 = 0; c0 < nsel - 1; c0 += 64) { // segment() would index an empty t_stats / wrap a negative slice
            const int c = c0 + lane;
            bool bad = false;
            if (c < nsel - 1) {
                const int lo = v.fend(start + c * sis) - 1 - 3;
                int hi = v.fend(start + (c + 1) * sis) - 1 + 3;
                if (hi > T) hi = T;
                bad = lo < 0 || hi - lo - 5 <= 0;
            }
            if (__ballot(bad)) status = WSX_READ_SEGMENT_RANGE;
        }
    }
    double cost = __builtin_nan("");
    if (status == 0) {
        int clo, chi;
        py_slice(start, end, na, &clo, &chi);
        if (chi > clo) {
            const int len = chi - clo;
            // (more than 2048 records: the stack lives in the read's slice of the per-sample scratch; every lane computes
            // the same sum, so only lane 0 may write there)
            double sum;
            if (len <= 128) sum = pw_block_wave(alc + clo, len, lane);
            else if (len <= 2048) sum = np_pairwise_sum(LoadPlain{alc + clo}, len, nullptr);
            else {
                sum = lane == 0 ? np_pairwise_sum(LoadPlain{alc + clo}, len, a.scr1 + v.off) : 0.0;
                sum = readlane_f64(sum, 0);
            }
            cost = sum / (double)len;
        }
        rec.start = start;
        rec.nsel = nsel;
        rec.p_lo = v.fend(start) - 1 - 3;
        int p_hi = v.fend(start + (nsel - 1) * sis) - 1 + 3;
        rec.p_hi = p_hi > T ? T : p_hi;
        rec.n_good = n_good;
    }
    if (lane != 0) return;
    a.rec[lr] = rec;
    a.status[lr] = status;
    res->status = status;
    if (a.pass == 1) {
        res->len1 = seqlen;
        res->n_trans1 = n;
        res->cost1 = cost;
        res->len2 = 0;
        res->n_trans2 = 0;
        res->reserved = 0;
        res->cost2 = __builtin_nan("");
        res->dtw_end_cost1 = a.end_cost ? a.end_cost[lr] : kInf;
        res->dtw_end_cost2 = kInf;
    } else {
        res->len2 = seqlen;
        res->n_trans2 = n;
        res->cost2 = cost;
        if (a.end_cost) res->dtw_end_cost2 = a.end_cost[lr];
    }
}

// (2,3,5) once more with one THREAD per read: per read it is a serial walk over the run list (every step a dependent
// load), but 64 reads share each instruction -- less total work than a wavefront per read, so big launches take this one
// (a launch lasts at least the ~50 us one thread needs).
__global__ __launch_bounds__(64) void borders_thread_kernel(MidArgs a)
{
    const int lr = blockIdx.x * blockDim.x + threadIdx.x;
    if (lr >= a.n_reads) return;
    wsx_result *res = (wsx_result *)a.results + lr;
    MidRec rec{};
    int status = a.status[lr];
    if (status != 0) {
        res->status = status;
        if (a.pass == 1) {
            res->len1 = res->len2 = res->n_trans1 = res->n_trans2 = 0;
            res->reserved = 0;
            res->cost1 = res->cost2 = __builtin_nan("");
            res->dtw_end_cost1 = res->dtw_end_cost2 = kInf;
        }
        a.rec[lr] = rec;
        return;
    }
    const ReadView v = view(a, lr);
    const DevAutomaton &A = a.aut[a.aut_id[v.r]];
    const int n = v.n, T = v.T, sis = a.prm.states_in_segment;
    const uint8_t *alg = a.al_good + v.off;
    const double *alc = a.al_cost + v.off;

    int slo, shi;
    py_slice((long long)A.flank_length - A.seq_idx[v.fstate(0)], -(long long)A.flank_length, n, &slo, &shi);
    const int seqlen = shi - slo;

    const int na = a.n_align ? a.n_align[lr] : n; // alignment records: one per run, or per distinct state
    int n_good = 0, start = -1, end = -1;
    for (int k = 0; k < na; k++) n_good += alg[k];
    for (int k = 0; k < n; k++) {
        if (A.repeat_mask[v.fstate(k)]) {
            if (start < 0) start = k;
            end = k;
        }
    }
    int nsel = 0, nb = 0;
    if (n_good < 4) {
        status = WSX_READ_FIT_POINTS; // rescale_signal runs (and fails) before mask_bad_repeats upstream
    } else if (start < 0) {
        status = WSX_READ_NO_REPEAT;
    } else {
        nb = end - start;
        const int add = (((nb - 1) % sis) + sis) % sis;
        if (add > 0) {
            end = end + (sis - add);
            if (end >= n) {
                status = WSX_READ_SEGMENT_RANGE;
            } else {
                const int es = v.fstate(end);
                int eR = end;
                for (int k = n - 1; k > end; k--)
                    if (v.fstate(k) == es) {
                        eR = k;
                        break;
                    }
                nb = eR - start;
            }
        }
        if (status == 0) {
            nsel = nb > 0 ? (nb + sis - 1) / sis : 0;
            if (nsel == 0) status = WSX_READ_SEGMENT_RANGE;
        }
    }
    if (status == 0) {
        for (int c = 0; c < nsel - 1; c++) { // segment() would index an empty t_stats / wrap a negative slice
            const int lo = v.fend(start + c * sis) - 1 - 3;
            int hi = v.fend(start + (c + 1) * sis) - 1 + 3;
            if (hi > T) hi = T;
            if (lo < 0 || hi - lo - 5 <= 0) status = WSX_READ_SEGMENT_RANGE;
        }
    }
    double cost = __builtin_nan("");
    if (status == 0) {
        int clo, chi;
        py_slice(start, end, na, &clo, &chi);
        if (chi > clo) cost = np_pairwise_sum(LoadPlain{alc + clo}, chi - clo, a.scr1 + v.off) / (double)(chi - clo);
        rec.start = start;
        rec.nsel = nsel;
        rec.p_lo = v.fend(start) - 1 - 3;
        int p_hi = v.fend(start + (nsel - 1) * sis) - 1 + 3;
        rec.p_hi = p_hi > T ? T : p_hi;
        rec.n_good = n_good;
    }
    a.rec[lr] = rec;
    a.status[lr] = status;
    res->status = status;
    if (a.pass == 1) {
        res->len1 = seqlen;
        res->n_trans1 = n;
        res->cost1 = cost;
        res->len2 = 0;
        res->n_trans2 = 0;
        res->reserved = 0;
        res->cost2 = __builtin_nan("");
        res->dtw_end_cost1 = a.end_cost ? a.end_cost[lr] : kInf;
        res->dtw_end_cost2 = kInf;
    } else {
        res->len2 = seqlen;
        res->n_trans2 = n;
        res->cost2 = cost;
        if (a.end_cost) res->dtw_end_cost2 = a.end_cost[lr];
    }
}

// (2b) optional: the called sequence as ASCII (WarpSTR._get_sequence, caller.py:178-187): last base of every visited
// state, flanks stripped with the reference's slice arithmetic, reverse-complemented for reverse-strand automata.
__global__ __launch_bounds__(256) void sequence_kernel(MidArgs a)
{
    const int lr = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (lr >= a.n_reads || a.status[lr] != 0) return;
    const ReadView v = view(a, lr);
    const DevAutomaton &A = a.aut[a.aut_id[v.r]];
    int slo, shi;
    py_slice((long long)A.flank_length - A.seq_idx[v.fstate(0)], -(long long)A.flank_length, v.n, &slo, &shi);
    uint8_t *out = a.seq_out + v.off;
    const int len = shi - slo;
    for (int q = lane; q < len; q += 64) {
        uint8_t b = A.last_base[v.fstate(slo + q)];
        int pos = q;
        if (A.reverse) {
            pos = len - 1 - q;
            b = b == 'A' ? 'T' : b == 'T' ? 'A' : b == 'C' ? 'G' : b == 'G' ? 'C' : b;
        }
        out[pos] = b;
    }
}

// (4a) segment() slides two 3-sample windows over a chunk; the t-statistic at absolute sample p only depends on
// sig[p-3..p+2], so it is computed once per position for the whole span of the chunks (calc_ttest, caller.py:347-354;
// np.std(..)**2 squares the rooted value, sic).  Every block also clears its tile of the output mask.
__global__ __launch_bounds__(256) void tstat_kernel(MidArgs a)
{
    __shared__ double wm[256 + 3], ws[256 + 3];
    const int lr = blockIdx.x;
    const ReadView v = view(a, lr);
    const int p0 = blockIdx.y * 256;
    if (p0 >= v.T) return;
    const int tid = threadIdx.x;
    if (a.maskbits && tid < 8 && p0 + tid * 32 < v.T) a.maskbits[v.off / 32 + lr + (p0 >> 5) + tid] = 0;
    if (a.badmask_bytes && p0 + tid < v.T) a.badmask_bytes[v.off + p0 + tid] = 0;
    if (a.status[lr] != 0) return;
    const MidRec rec = a.rec[lr];
    if (rec.nsel <= 1 || p0 + 256 <= rec.p_lo || p0 >= rec.p_hi) return;
    const double *sig = a.signal + v.off;
    const double r3 = wsx_recip_refined(3.0);
    // window statistics for positions p0-3 .. p0+255 (those inside [p_lo, p_hi-2))
    for (int q = tid; q < 259; q += 256) {
        const int p = p0 - 3 + q;
        double mu = 0.0, ss = 0.0;
        if (p >= rec.p_lo && p + 2 < rec.p_hi) {
            const double a0 = sig[p], a1 = sig[p + 1], a2 = sig[p + 2];
            mu = mean3v(a0, a1, a2, r3);
            const double sd = std3v(a0, a1, a2, mu, r3);
            ss = sd * sd;
        }
        wm[q] = mu;
        ws[q] = ss;
    }
    __syncthreads();
    const int p = p0 + tid;
    if (p >= rec.p_lo + 3 && p + 2 < rec.p_hi) {
        double sd = sqrt(wsx_div_by(ws[tid] + ws[tid + 3], 3.0, r3));
        if (sd == 0.0) sd = sd + 0.0000001;
        a.scr2[v.off + p] = (wm[tid] - wm[tid + 3]) / sd;
    }
}

// (4b) segment() peak scan per chunk (caller.py:357-378), check_segments (342-344), mask_big_events (409-421)
__global__ __launch_bounds__(64) void chunk_kernel(MidArgs a)
{
    const int lr = blockIdx.x;
    if (a.status[lr] != 0) return;
    const MidRec rec = a.rec[lr];
    const int c = blockIdx.y * blockDim.x + threadIdx.x;
    if (c >= rec.nsel - 1) return;
    const ReadView v = view(a, lr);
    const int sis = a.prm.states_in_segment;
    const int b0 = v.fend(rec.start + c * sis) - 1, b1 = v.fend(rec.start + (c + 1) * sis) - 1;
    int hi = b1 + 3;
    if (hi > v.T) hi = v.T;
    const double *tst = a.scr2 + v.off;
    int borders = 0;
    bool started = false;
    double prev = tst[b0];
    for (int p = b0; p + 2 < hi; p++) {
        const double t = tst[p];
        if (t > 3 || t < -3) {
            if ((t > 3 && t >= prev) || (t < -3 && t <= prev)) {
                started = true;
            } else {
                if (started) borders++;
                started = false;
            }
        } else if (started) {
            borders++;
            started = false;
        }
        prev = t;
    }
    if (borders - 1 >= sis + 1) { // check_segments: >= states_in_segment + 1
        uint32_t *mw = a.maskbits + (v.off / 32 + lr);
        for (int w = b0 >> 5; w <= ((b1 - 1) >> 5); w++) {
            const int q0 = w * 32 > b0 ? w * 32 : b0, q1 = (w + 1) * 32 < b1 ? (w + 1) * 32 : b1;
            const uint32_t bits = (q1 - q0 >= 32) ? 0xffffffffu : (((1u << (q1 - q0)) - 1u) << (q0 & 31));
            atomicOr(&mw[w], bits);
        }
        if (a.badmask_bytes)
            for (int q = b0; q < b1; q++) a.badmask_bytes[v.off + q] = 1;
    }
}

// (4) = (4a) + (4b) in one pass, one 256-thread block per read: the t-statistics stay in LDS, and the peak scan of
// segment() becomes a count.  The scan's state machine is local: with A(p) := "t[p] is beyond +-3 and not falling back"
// ((t > 3 and t >= prev) or (t < -3 and t <= prev), prev = t[p-1], or t[p] itself at the first position of the chunk),
// `started` before position p is exactly A(p-1), so  borders(chunk) = #{p in (b0, b1] : A(p-1) and not A(p)}.
__global__ __launch_bounds__(256) void segment_kernel(MidArgs a, int max_chunks)
{
    extern __shared__ int seg_lds[]; // cb[max_chunks + 1]: first sample of every chunk; cnt[max_chunks]
    __shared__ double wm[256 + 3], ws[256 + 3], tl[256 + 2];
    int *cb = seg_lds, *cnt = seg_lds + (max_chunks + 1);
    const int lr = blockIdx.x, tid = threadIdx.x;
    const ReadView v = view(a, lr);
    uint32_t *mw = a.maskbits + (v.off / 32 + lr);
    for (int w = tid; w < (v.T + 31) / 32; w += 256) mw[w] = 0;
    if (a.badmask_bytes)
        for (int q = tid; q < v.T; q += 256) a.badmask_bytes[v.off + q] = 0;
    if (a.status[lr] != 0) return;
    const MidRec rec = a.rec[lr];
    const int nch = rec.nsel - 1, sis = a.prm.states_in_segment;
    if (nch < 1 || nch + 1 > max_chunks) return;
    for (int c = tid; c <= nch; c += 256) {
        cb[c] = v.fend(rec.start + c * sis) - 1;
        if (c < nch) cnt[c] = 0;
    }
    const double *sig = a.signal + v.off;
    const double r3 = wsx_recip_refined(3.0);
    for (int p0 = rec.p_lo + 3; p0 + 2 < rec.p_hi; p0 += 256) { // positions p0 .. p0+255 (t exists on [p_lo+3, p_hi-2))
        const double carry = (tid < 2 && p0 > rec.p_lo + 3) ? tl[256 + tid] : 0.0; // t[p0-2], t[p0-1] of the last tile
        for (int q = tid; q < 259; q += 256) { // window statistics at p0-3 .. p0+255
            const int p = p0 - 3 + q;
            double mu = 0.0, ss = 0.0;
            if (p >= rec.p_lo && p + 2 < rec.p_hi) {
                const double a0 = sig[p], a1 = sig[p + 1], a2 = sig[p + 2];
                mu = mean3v(a0, a1, a2, r3);
                const double sd = std3v(a0, a1, a2, mu, r3);
                ss = sd * sd;
            }
            wm[q] = mu;
            ws[q] = ss;
        }
        __syncthreads();
        const int p = p0 + tid;
        const bool valid = p + 2 < rec.p_hi;
        double t = 0.0;
        if (valid) {
            double sd = sqrt(wsx_div_by(ws[tid] + ws[tid + 3], 3.0, r3));
            if (sd == 0.0) sd = sd + 0.0000001;
            t = (wm[tid] - wm[tid + 3]) / sd;
        }
        tl[2 + tid] = t;
        if (tid < 2) tl[tid] = carry;
        __syncthreads();
        if (valid && p > cb[0]) {
            int lo = 0, hi = nch - 1; // the chunk whose first sample is the last one before p
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (cb[mid] < p) lo = mid;
                else hi = mid - 1;
            }
            const double t1 = tl[1 + tid], prev1 = (p - 1 == cb[lo]) ? t1 : tl[tid];
            const bool a1 = (t1 > 3 && t1 >= prev1) || (t1 < -3 && t1 <= prev1);
            const bool a0 = (t > 3 && t >= t1) || (t < -3 && t <= t1);
            if (a1 && !a0) atomicAdd(&cnt[lo], 1);
        }
    }
    __syncthreads();
    for (int c = tid; c < nch; c += 256) {
        if (cnt[c] - 1 >= sis + 1) { // check_segments: >= states_in_segment + 1
            const int b0 = cb[c], b1 = cb[c + 1];
            for (int w = b0 >> 5; w <= ((b1 - 1) >> 5); w++) {
                const int q0 = w * 32 > b0 ? w * 32 : b0, q1 = (w + 1) * 32 < b1 ? (w + 1) * 32 : b1;
                const uint32_t bits = (q1 - q0 >= 32) ? 0xffffffffu : (((1u << (q1 - q0)) - 1u) << (q0 & 31));
                atomicOr(&mw[w], bits);
            }
            if (a.badmask_bytes)
                for (int q = b0; q < b1; q++) a.badmask_bytes[v.off + q] = 1;
        }
    }
}

// (6) rescaling input: accepted records, stably sorted by value (filter_alignment + list.sort, caller.py:304-318):
// compact (order preserved), then rank = #(x_q < x_k) + #(x_q == x_k, q < k) -- counted inside value buckets for the usual
// 100-1000 records, directly for fewer or more.  One wavefront per read.
__global__ __launch_bounds__(64) void sort_kernel(MidArgs a)
{
    constexpr int SORT_CAP = 512, SORT_BUCKETS = 512, SORT_DIRECT_MAX = 96; // (fewer than ~100 records: counting is cheaper)
    __shared__ double xs[SORT_CAP]; // direct counting: a tile of the values; bucketed: the values grouped by bucket
    __shared__ int sort_cnt[SORT_BUCKETS], sort_start[SORT_BUCKETS + 1];
    __shared__ uint32_t sort_bs[SORT_CAP];  // per element: bucket | slot inside the bucket << 16
    __shared__ uint16_t sort_idx[SORT_CAP]; // the elements' positions, grouped like xs
    const int lane = threadIdx.x & 63;
    const int lr = blockIdx.x;
    if (a.status[lr] != 0) return;
    const ReadView v = view(a, lr);
    const int n = a.n_align ? a.n_align[lr] : v.n;
    const double *alv = a.al_value + v.off, *ale = a.al_expected + v.off;
    const uint8_t *alg = a.al_good + v.off;
    double *cx = a.scr0 + v.off, *cy = a.scr1 + v.off;
    double *fx = a.fit_x + v.off, *fy = a.fit_y + v.off;
    int base = 0;
    for (int kb = 0; kb < n; kb += 64) {
        const int k = kb + lane;
        const bool g = (k < n) && alg[k];
        const unsigned long long bal = __ballot(g);
        const int pos = base + __builtin_popcountll(bal & ((1ull << lane) - 1ull));
        if (g) {
            cx[pos] = alv[k];
            cy[pos] = ale[k];
        }
        base += __builtin_popcountll(bal);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const int mfit = base;
    if (mfit > SORT_DIRECT_MAX && mfit <= SORT_CAP) {
        // Bucketed rank.  b(x) = int((x - min) * (SORT_BUCKETS - 1) / (max - min)) never decreases with x and is the same
        // for equal values, so an element's rank is the number of elements in lower buckets plus its rank among the members
        // of its own bucket -- a handful of comparisons instead of mfit.  The order is the same total order as below (value,
        // then position), whatever the bucket boundaries are.
        double mn = kInf, mx = -kInf;
        for (int k = lane; k < mfit; k += 64) {
            const double x = cx[k];
            mn = x < mn ? x : mn;
            mx = x > mx ? x : mx;
        }
        for (int o = 32; o > 0; o >>= 1) {
            const double m1 = __shfl_xor(mn, o), m2 = __shfl_xor(mx, o);
            mn = m1 < mn ? m1 : mn;
            mx = m2 > mx ? m2 : mx;
        }
        const double scale = mx > mn ? (double)(SORT_BUCKETS - 1) / (mx - mn) : 0.0;
        auto bucket = [&](double x) {
            const int b = (int)((x - mn) * scale);
            return b < 0 ? 0 : (b > SORT_BUCKETS - 1 ? SORT_BUCKETS - 1 : b);
        };
        for (int b = lane; b < SORT_BUCKETS; b += 64) sort_cnt[b] = 0;
        __builtin_amdgcn_wave_barrier();
        for (int k = lane; k < mfit; k += 64) {
            const int b = bucket(cx[k]);
            const int slot = atomicAdd(&sort_cnt[b], 1);
            sort_bs[k] = (uint32_t)b | ((uint32_t)slot << 16);
        }
        __builtin_amdgcn_wave_barrier();
        { // exclusive prefix sum of the bucket counts: SORT_BUCKETS / 64 consecutive buckets per lane, then across the lanes
            constexpr int PER = SORT_BUCKETS / 64;
            int c[PER], tot = 0;
#pragma unroll
            for (int q = 0; q < PER; q++) {
                c[q] = sort_cnt[lane * PER + q];
                tot += c[q];
            }
            int incl = tot;
            for (int o = 1; o < 64; o <<= 1) {
                const int up = __shfl_up(incl, o);
                if (lane >= o) incl += up;
            }
            int run = incl - tot;
#pragma unroll
            for (int q = 0; q < PER; q++) {
                sort_start[lane * PER + q] = run;
                run += c[q];
            }
            if (lane == 63) sort_start[SORT_BUCKETS] = run;
        }
        __builtin_amdgcn_wave_barrier();
        for (int k = lane; k < mfit; k += 64) {
            const uint32_t p = sort_bs[k];
            const int at = sort_start[p & 0xffffu] + (int)(p >> 16);
            xs[at] = cx[k];
            sort_idx[at] = (uint16_t)k;
        }
        __builtin_amdgcn_wave_barrier();
        for (int kb = 0; kb < mfit; kb += 64) {
            const int k = kb + lane;
            const bool valid = k < mfit;
            const double xk = valid ? cx[k] : 0.0;
            const int b = valid ? (int)(sort_bs[k] & 0xffffu) : 0;
            const int j0 = sort_start[b], j1 = valid ? sort_start[b + 1] : j0;
            int below = 0;
            for (int j = j0; __any(j < j1); j++)
                if (j < j1) {
                    const double xq = xs[j];
                    below += (xq < xk || (xq == xk && (int)sort_idx[j] < k)) ? 1 : 0;
                }
            if (valid) {
                fx[j0 + below] = xk;
                fy[j0 + below] = cy[k];
            }
        }
        if (lane == 0) a.fit_m[lr] = mfit;
        return;
    }
    for (int kb = 0; kb < mfit; kb += 64) {
        const int k = kb + lane;
        const double xk = (k < mfit) ? cx[k] : 0.0;
        // rank = #(x_q < x_k) + #(x_q == x_k, q < k).  Both k and q advance in aligned blocks of 64, so a block of q
        // lies entirely before the lanes' k (count "<="), entirely after (count "<"), or is the lanes' own block (the
        // index decides ties): one compare and one add-with-carry per (q, k) pair except on the diagonal.
        int rank = 0;
        for (int qb = 0; qb < mfit; qb += 512) {
            const int cnt = (mfit - qb) < 512 ? (mfit - qb) : 512;
            __builtin_amdgcn_wave_barrier();
            for (int q = lane; q < cnt; q += 64) xs[q] = cx[qb + q];
            __builtin_amdgcn_wave_barrier();
            for (int q0 = 0; q0 < cnt; q0 += 64) {
                const int q1 = (q0 + 64 < cnt) ? q0 + 64 : cnt;
                if (qb + q0 < kb) {
                    for (int q = q0; q < q1; q++) rank += (xs[q] <= xk) ? 1 : 0; // LDS broadcast
                } else if (qb + q0 > kb) {
                    for (int q = q0; q < q1; q++) rank += (xs[q] < xk) ? 1 : 0;
                } else {
                    for (int q = q0; q < q1; q++) {
                        const double xq = xs[q];
                        rank += (xq < xk || (xq == xk && q - q0 < lane)) ? 1 : 0;
                    }
                }
            }
        }
        if (k < mfit) {
            fx[rank] = xk;
            fy[rank] = cy[k];
        }
    }
    if (lane == 0) a.fit_m[lr] = mfit;
}

// ---- FITPACK pieces ---------------------------------------------------------------------------
// fpbspl for the 8-knot cubic (knots xb x4, xe x4; interval l = 4): the 4 non-zero B-splines at x.
// den = xe - xb is the same for every point of a read; rden = wsx_recip_refined(den), or 0 where den is so close to the
// ends of the exponent range that the plain division might scale its operands (wsx_bspl_rden).
__device__ __forceinline__ double wsx_bspl_rden(double den)
{
    return (den > 0x1p-100 && den < 0x1p100) ? wsx_recip_refined(den) : 0.0;
}
__device__ __forceinline__ void bspl4(double xb, double xe, double den, double rden, double x, double h[5])
{
    double hh[4];
    h[1] = 1.0;
#pragma unroll
    for (int j = 1; j <= 3; j++) {
#pragma unroll
        for (int i = 1; i <= j; i++) hh[i] = h[i];
        h[1] = 0.0;
#pragma unroll
        for (int i = 1; i <= j; i++) {
            // t(l+i) = xe, t(l+i-j) = xb for l = 4
            const double f = rden != 0.0 ? wsx_div_by(hh[i], den, rden) : hh[i] / den;
            h[i] = h[i] + f * (xe - x);
            h[i + 1] = f * (x - xb);
        }
    }
}

__device__ __forceinline__ void givens(double piv, double &ww, double &c, double &s)
{
    const double store = fabs(piv);
    double dd;
    if (store >= ww) {
        const double q = ww / piv;
        dd = store * sqrt(1.0 + q * q);
    } else {
        const double q = piv / ww;
        dd = ww * sqrt(1.0 + q * q);
    }
    c = ww / dd;
    s = piv / dd;
    ww = dd;
}

__device__ __forceinline__ void rota(double c, double s, double &x, double &y)
{
    const double stor1 = x, stor2 = y;
    y = c * stor2 + s * stor1;
    x = c * stor1 - s * stor2;
}

// A double from another lane of the same quad (DPP quad_perm: no LDS, two 32-bit moves).
template <int CTRL>
__device__ __forceinline__ double quad_f64(double x)
{
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
constexpr int kQuadFromLeft = 0x90; // quad_perm [0,0,1,2]: lane j takes lane j-1's value (lane 0 keeps its own)
#define WSX_QUAD_BCAST(k) ((k) * 0x55)

// fpcurf (iopt = 0, k = 3, s = m, unit weights) restricted to its first iteration, + fpback.
// The observation rows go through the 4x4 banded triangle one after the other, and a row is rotated into triangle row
// 1, then 2, 3, 4 -- but rotating into row j touches nothing except row j of the triangle and the observation row itself.
// So the four rotations are a PIPELINE: lane j of a quad owns triangle row j+1 (diagonal, up to three off-diagonals,
// right-hand side) and at step t rotates point t-j into it, then hands the point's remaining entries to lane j+1 (DPP).
// Every value sees the same operations in the same order as in the one-thread recurrence (bit-identical results); the
// dependent chain per point shrinks from four Givens set-ups (each a division, a square root and two more divisions) to
// one, and a read takes 4 lanes instead of 1 thread, so small launches spread over four times as many wavefronts.
__global__ __launch_bounds__(64) void fit_kernel(FitArgs a)
{
    const int j = threadIdx.x & 3;
    const int lr = blockIdx.x * 16 + (threadIdx.x >> 2);
    if (lr >= a.n_reads) return; // whole quads leave together
    if (a.status[lr] != 0) return;
    const int r = a.first_read + lr;
    const long long off = a.offsets[r] - a.base_off;
    const int m = a.fit_m[lr];
    const double *x = a.fit_x + off, *y = a.fit_y + off;
    const double xb = x[0], xe = x[m - 1];
    if (!(xb < xe)) {
        if (j == 0) a.status[lr] = WSX_READ_FIT_ORDER;
        return;
    }
    const double den = xe - xb, rden = wsx_bspl_rden(den);
    // this lane's triangle row: diagonal, off-diagonals (absent ones stay 0: rotating zeros gives zeros), right-hand side
    double dg = 0, o0 = 0, o1 = 0, o2 = 0, z = 0, fp = 0.0;
    // what the lane to the left handed over at the end of the previous step
    double piv_in = 0, h0_in = 0, h1_in = 0, h2_in = 0, yi_in = 0;
    double xn = x[0], yn = y[0]; // the next point is fetched while the current one is in the pipeline
    for (int t = 0; t < m + 3; t++) {
        const double xc = xn, yc = yn;
        if (t + 1 < m) {
            xn = x[t + 1];
            yn = y[t + 1];
        }
        double h[5];
        bspl4(xb, xe, den, rden, xc, h); // (used by lane 0 only)
        double piv = piv_in, h0 = h0_in, h1 = h1_in, h2 = h2_in, yi = yi_in;
        if (j == 0) {
            yi = yc * 1.0; // unit weights
            piv = h[1] * 1.0;
            h0 = h[2] * 1.0;
            h1 = h[3] * 1.0;
            h2 = h[4] * 1.0;
        }
        const bool valid = t - j >= 0 && t - j < m;
        if (valid && piv != 0.0) {
            double c, s;
            givens(piv, dg, c, s);
            rota(c, s, yi, z);
            rota(c, s, h0, o0);
            rota(c, s, h1, o1);
            rota(c, s, h2, o2);
        }
        if (valid && j == 3) fp = fp + yi * yi;
        piv_in = quad_f64<kQuadFromLeft>(h0);
        h0_in = quad_f64<kQuadFromLeft>(h1);
        h1_in = quad_f64<kQuadFromLeft>(h2);
        h2_in = 0.0;
        yi_in = quad_f64<kQuadFromLeft>(yi);
    }
    fp = quad_f64<WSX_QUAD_BCAST(3)>(fp);
    // fpcurf's test after the first least-squares fit (s = m, acc = tol * s, tol = 0.001): |fp - s| < acc or fp < s keeps the
    // polynomial (ier = -2); anything else adds knots and smooths: fit_smooth_kernel.  With rescaling.threshold <= 1 this
    // cannot happen (the line y = x alone has residual <= m * threshold^2); above 1 it takes accepted states that are, on
    // average, further than one normalised unit from their level.
    if (!(fp - (double)m < 0.001 * (double)m)) {
        if (j == 0) {
            if (a.smooth_list) { // (handles with rescaling.threshold > 1) the read goes to fit_smooth_kernel
                const int slot = atomicAdd(a.smooth_count, 1);
                atomicMax(a.smooth_count + 1, m);
                a.smooth_list[slot] = lr;
                a.smooth_slot[lr] = slot;
            } else {
                a.status[lr] = WSX_READ_FIT_SMOOTH;
            }
        }
        return;
    }
    if (j == 0 && a.smooth_slot) a.smooth_slot[lr] = -1;
    // fpback (n = 4, bandwidth 4): every lane of the quad evaluates it on the gathered triangle
    const double A11 = quad_f64<WSX_QUAD_BCAST(0)>(dg), A12 = quad_f64<WSX_QUAD_BCAST(0)>(o0),
                 A13 = quad_f64<WSX_QUAD_BCAST(0)>(o1), A14 = quad_f64<WSX_QUAD_BCAST(0)>(o2),
                 z1 = quad_f64<WSX_QUAD_BCAST(0)>(z);
    const double A21 = quad_f64<WSX_QUAD_BCAST(1)>(dg), A22 = quad_f64<WSX_QUAD_BCAST(1)>(o0),
                 A23 = quad_f64<WSX_QUAD_BCAST(1)>(o1), z2 = quad_f64<WSX_QUAD_BCAST(1)>(z);
    const double A31 = quad_f64<WSX_QUAD_BCAST(2)>(dg), A32 = quad_f64<WSX_QUAD_BCAST(2)>(o0),
                 z3 = quad_f64<WSX_QUAD_BCAST(2)>(z);
    const double A41 = quad_f64<WSX_QUAD_BCAST(3)>(dg), z4 = quad_f64<WSX_QUAD_BCAST(3)>(z);
    const double c4 = z4 / A41;
    const double c3 = (z3 - c4 * A32) / A31;
    double st = z2;
    st = st - c3 * A22;
    st = st - c4 * A23;
    const double c2 = st / A21;
    st = z1;
    st = st - c2 * A12;
    st = st - c3 * A13;
    st = st - c4 * A14;
    const double c1 = st / A11;
    if (j != 0) return;
    double *co = a.coef + (size_t)lr * 6;
    co[0] = xb;
    co[1] = xe;
    co[2] = c1;
    co[3] = c2;
    co[4] = c3;
    co[5] = c4;
}

// ---- FITPACK beyond the polynomial: fpcurf in full (rescaling.threshold > 1 only) ----------------------------------
// One THREAD per read that failed fpcurf's first test; the read's arrays live in its slot of the handle's smoothing
// workspace (global memory, 1-based like the published routines).  Operation for operation the oracle's wso_curfit,
// which is pinned bit for bit against SciPy's compiled FITPACK.  A rare path: what matters is that it is exact.
// Slot layout (doubles): [0] n (as int), then t, c, z, fpint (nest + 2 each), nrdata (ints), a (x5), b (x6), g (x6), q (x5).
__host__ __device__ inline size_t smooth_slot_doubles(int nest) { return (size_t)27 * (nest + 2) + 8; }

__device__ __forceinline__ void bspl_knots(const double *t, double x, int l, double h[6])
{
    double hh[4];
    h[1] = 1.0;
#pragma unroll
    for (int j = 1; j <= 3; j++) {
#pragma unroll
        for (int i = 1; i <= j; i++) hh[i] = h[i];
        h[1] = 0.0;
#pragma unroll
        for (int i = 1; i <= j; i++) {
            const double tli = t[l + i], tlj = t[l + i - j];
            if (tli == tlj) {
                h[i + 1] = 0.0;
            } else {
                const double f = hh[i] / (tli - tlj);
                h[i] = h[i] + f * (tli - x);
                h[i + 1] = f * (x - tlj);
            }
        }
    }
}

// fpback: a*c = z, a upper triangular with bandwidth k, rows of `a` `ld` doubles apart
__device__ __forceinline__ void back_subst(const double *a, int ld, const double *z, int n, int k, double *c)
{
    const int k1 = k - 1;
    c[n] = z[n] / a[n * ld + 1];
    int i = n - 1;
    if (i == 0) return;
    for (int j = 2; j <= n; j++) {
        double store = z[i];
        int i1 = k1;
        if (j <= k1) i1 = j - 1;
        int mm = i;
        for (int l = 1; l <= i1; l++) {
            mm = mm + 1;
            store = store - c[mm] * a[i * ld + l + 1];
        }
        c[i] = store / a[i * ld + 1];
        i = i - 1;
    }
}

__global__ __launch_bounds__(64) void fit_smooth_kernel(FitArgs a, int count, int nest_cap, double *ws)
{
    const int slot = blockIdx.x * 64 + threadIdx.x;
    if (slot >= count) return;
    const int lr = a.smooth_list[slot];
    const int r = a.first_read + lr;
    const long long off = a.offsets[r] - a.base_off;
    const int m = a.fit_m[lr];
    const double *x = a.fit_x + off - 1, *y = a.fit_y + off - 1; // 1-based
    constexpr int K = 3, K1 = 4, K2 = 5, MAXIT = 20;
    const double s = (double)m, tol = 0.001, con1 = 0.1, con9 = 0.9, con4 = 0.04, half = 0.5;
    const int nest = m + K1 > 2 * K + 3 ? m + K1 : 2 * K + 3;
    double *base = ws + (size_t)slot * smooth_slot_doubles(nest_cap);
    const size_t nd = (size_t)nest_cap + 2;
    double *t = base + 8, *c = t + nd, *z = c + nd, *fpint = z + nd;
    int *nrdata = (int *)(fpint + nd);
    double *wa = fpint + nd + nd / 2 + 1, *wb = wa + nd * 5, *wg = wb + nd * 6, *wq = wg + nd * 6;
#define A_(i, j) wa[(i) * 5 + (j)]
#define B_(i, j) wb[(i) * 6 + (j)]
#define G_(i, j) wg[(i) * 6 + (j)]
#define Q_(i, j) wq[(i) * 5 + (j)]
    const double xb = x[1], xe = x[m];
    const int nmin = 2 * K1, nmax = m + K1;
    const double acc = tol * s;
    int n = nmin, ier = 0, nplus = 0, nrint = 0, nk1 = 0;
    double fp = 0.0, fp0 = 0.0, fpold = 0.0, fpms = 0.0;
    nrdata[1] = m - 2;
    bool accepted = false, smoothing = false;
    for (int entry = 0; entry < 2 && !accepted && !smoothing; entry++) { // (second entry: the interpolation knots)
        bool again = false;
        for (int iter = 1; iter <= m; iter++) {
            if (n == nmin) ier = -2;
            nrint = n - nmin + 1;
            nk1 = n - K1;
            {
                int i = n;
                for (int j = 1; j <= K1; j++) {
                    t[j] = xb;
                    t[i] = xe;
                    i = i - 1;
                }
            }
            fp = 0.0;
            for (int i = 1; i <= nk1; i++) {
                z[i] = 0.0;
                for (int j = 1; j <= K1; j++) A_(i, j) = 0.0;
            }
            int l = K1;
            for (int it = 1; it <= m; it++) {
                const double xi = x[it];
                double yi = y[it] * 1.0;
                while (!(xi < t[l + 1] || l == nk1)) l = l + 1;
                double h[6];
                bspl_knots(t, xi, l, h);
#pragma unroll
                for (int i = 1; i <= K1; i++) {
                    Q_(it, i) = h[i];
                    h[i] = h[i] * 1.0;
                }
                bool last = false;
#pragma unroll
                for (int i = 1; i <= K1; i++) {
                    const int j = l - K1 + i;
                    const double piv = h[i];
                    if (!last && piv != 0.0) {
                        double cs, sn;
                        givens(piv, A_(j, 1), cs, sn);
                        rota(cs, sn, yi, z[j]);
                        if (i == K1) {
                            last = true;
                        } else {
#pragma unroll
                            for (int i1 = i + 1; i1 <= K1; i1++) rota(cs, sn, h[i1], A_(j, i1 - i + 1));
                        }
                    }
                }
                fp = fp + yi * yi;
            }
            if (ier == -2) fp0 = fp;
            fpint[n] = fp0;
            fpint[n - 1] = fpold;
            nrdata[n] = nplus;
            back_subst(wa, 5, z, nk1, K1, c);
            fpms = fp - s;
            if (fabs(fpms) < acc) {
                accepted = true;
                break;
            }
            if (fpms < 0.0) {
                smoothing = true;
                break;
            }
            if (n == nmax) {
                ier = -1;
                accepted = true;
                break;
            }
            if (n == nest) {
                ier = 1;
                accepted = true;
                break;
            }
            if (ier != 0) {
                nplus = 1;
                ier = 0;
            } else {
                int npl1 = nplus * 2;
                const double rn = (double)nplus;
                if (fpold - fp > acc) { // (out of the integer range: INT_MIN, as the host's conversion gives and the oracle restates)
                    const double v = rn * fpms / (fpold - fp);
                    npl1 = (v > -2147483649.0 && v < 2147483648.0) ? (int)v : (-2147483647 - 1);
                }
                int mx = npl1 > nplus / 2 ? npl1 : nplus / 2;
                if (mx < 1) mx = 1;
                nplus = nplus * 2 < mx ? nplus * 2 : mx;
            }
            fpold = fp;
            double fpart = 0.0;
            int i = 1, nw = 0;
            l = K2;
            for (int it = 1; it <= m; it++) {
                if (!(x[it] < t[l] || l > nk1)) {
                    nw = 1;
                    l = l + 1;
                }
                double term = 0.0;
#pragma unroll
                for (int j = 1; j <= K1; j++) term = term + c[l - K2 + j] * Q_(it, j);
                term = (1.0 * (term - y[it])) * (1.0 * (term - y[it]));
                fpart = fpart + term;
                if (nw == 0) continue;
                const double store = term * half;
                fpint[i] = fpart - store;
                i = i + 1;
                fpart = store;
                nw = 0;
            }
            fpint[nrint] = fpart;
            bool to_interp = false;
            for (l = 1; l <= nplus; l++) {
                // fpknot: one more knot in the interval with the largest residual that still holds data points
                {
                    const int k = (n - nrint - 1) / 2;
                    double fpmax = 0.0;
                    int jbegin = 1, number = 0, maxpt = 0, maxbeg = 0;
                    for (int j = 1; j <= nrint; j++) {
                        const int jpoint = nrdata[j];
                        if (!(fpmax >= fpint[j] || jpoint == 0)) {
                            fpmax = fpint[j];
                            number = j;
                            maxpt = jpoint;
                            maxbeg = jbegin;
                        }
                        jbegin = jbegin + jpoint + 1;
                    }
                    if (number != 0) {
                        const int ihalf = maxpt / 2 + 1, nrx = maxbeg + ihalf, next = number + 1;
                        for (int j = next; j <= nrint; j++) {
                            const int jj = next + nrint - j;
                            fpint[jj + 1] = fpint[jj];
                            nrdata[jj + 1] = nrdata[jj];
                            t[jj + k + 1] = t[jj + k];
                        }
                        nrdata[number] = ihalf - 1;
                        nrdata[next] = maxpt - ihalf;
                        const double am = (double)maxpt;
                        double an = (double)nrdata[number];
                        fpint[number] = fpmax * an / am;
                        an = (double)nrdata[next];
                        fpint[next] = fpmax * an / am;
                        t[next + k] = x[nrx];
                        n = n + 1;
                        nrint = nrint + 1;
                    }
                }
                if (n == nmax) {
                    to_interp = true;
                    break;
                }
                if (n == nest) break;
            }
            if (to_interp) {
                int ii = K2, jj = K / 2 + 2;
                for (l = 1; l <= m - K1; l++) {
                    t[ii] = x[jj];
                    ii = ii + 1;
                    jj = jj + 1;
                }
                again = true;
                break;
            }
        }
        if (!again) break;
    }
    if (!accepted && ier != -2) {
        // part 2: the smoothing spline
        {   // fpdisc
            const int nrint2 = nk1 - K;
            const double an = (double)nrint2;
            const double fac = an / (t[nk1 + 1] - t[K1]);
            for (int l = K2; l <= nk1; l++) {
                const int lmk = l - K1;
                double h[13];
#pragma unroll
                for (int j = 1; j <= K1; j++) {
                    h[j] = t[l] - t[l + j - K2];
                    h[j + K1] = t[l] - t[l + j];
                }
#pragma unroll
                for (int j = 1; j <= K2; j++) {
                    double prod = h[j];
#pragma unroll
                    for (int i = 1; i <= K; i++) prod = prod * h[j + i] * fac;
                    const int lp = lmk + j - 1;
                    B_(lmk, j) = (t[lp + K1] - t[lp]) / prod;
                }
            }
        }
        double p1 = 0.0, f1 = fp0 - s, p3 = -1.0, f3 = fpms, p = 0.0;
        for (int i = 1; i <= nk1; i++) p = p + A_(i, 1);
        p = (double)nk1 / p;
        int ich1 = 0, ich3 = 0;
        const int n8 = n - nmin;
        for (int iter = 1; iter <= MAXIT; iter++) {
            const double pinv = 1.0 / p;
            for (int i = 1; i <= nk1; i++) {
                c[i] = z[i];
                G_(i, K2) = 0.0;
                for (int j = 1; j <= K1; j++) G_(i, j) = A_(i, j);
            }
            for (int it = 1; it <= n8; it++) {
                double h[7];
#pragma unroll
                for (int i = 1; i <= K2; i++) h[i] = B_(it, i) * pinv;
                h[6] = 0.0;
                double yi = 0.0;
                for (int j = it; j <= nk1; j++) {
                    const double piv = h[1];
                    double cs, sn;
                    givens(piv, G_(j, 1), cs, sn);
                    rota(cs, sn, yi, c[j]);
                    if (j == nk1) break;
                    int i2 = K1;
                    if (j > n8) i2 = nk1 - j;
#pragma unroll
                    for (int i = 1; i <= K1; i++)
                        if (i <= i2) {
                            rota(cs, sn, h[i + 1], G_(j, i + 1));
                            h[i] = h[i + 1];
                        }
#pragma unroll
                    for (int i = 1; i <= K2; i++)
                        if (i == i2 + 1) h[i] = 0.0;
                }
            }
            back_subst(wg, 6, c, nk1, K2, c);
            fp = 0.0;
            int l = K2;
            for (int it = 1; it <= m; it++) {
                if (!(x[it] < t[l] || l > nk1)) l = l + 1;
                double term = 0.0;
#pragma unroll
                for (int j = 1; j <= K1; j++) term = term + c[l - K2 + j] * Q_(it, j);
                fp = fp + (1.0 * (term - y[it])) * (1.0 * (term - y[it]));
            }
            fpms = fp - s;
            if (fabs(fpms) < acc) break;
            if (iter == MAXIT) {
                ier = 3;
                break;
            }
            const double p2 = p, f2 = fpms;
            if (ich3 == 0) {
                if (!(f2 - f3 > acc)) {
                    p3 = p2;
                    f3 = f2;
                    p = p * con4;
                    if (p <= p1) p = p1 * con9 + p2 * con1;
                    continue;
                }
                if (f2 < 0.0) ich3 = 1;
            }
            if (ich1 == 0) {
                if (!(f1 - f2 > acc)) {
                    p1 = p2;
                    f1 = f2;
                    p = p / con4;
                    if (p3 < 0.0) continue;
                    if (p >= p3) p = p2 * con1 + p3 * con9;
                    continue;
                }
                if (f2 > 0.0) ich1 = 1;
            }
            if (f2 >= f1 || f2 <= f3) {
                ier = 2;
                break;
            }
            // fprati
            if (p3 > 0.0) {
                const double h1 = f1 * (f2 - f3), h2 = f2 * (f3 - f1), h3 = f3 * (f1 - f2);
                p = -(p1 * p2 * h3 + p2 * p3 * h1 + p3 * p1 * h2) / (p1 * h1 + p2 * h2 + p3 * h3);
            } else {
                p = (p1 * (f1 - f3) * f2 - p2 * (f2 - f3) * f1) / ((f1 - f2) * f3);
            }
            if (f2 < 0.0) {
                p3 = p2;
                f3 = f2;
            } else {
                p1 = p2;
                f1 = f2;
            }
        }
    }
#undef A_
#undef B_
#undef G_
#undef Q_
    // a spline with a coefficient that is not finite (coinciding abscissae can do that) is of no use to the second pass
    bool finite = true;
    for (int i = 1; i <= n - K1; i++) finite = finite && (fabs(c[i]) <= 1.7976931348623157e308);
    *(int *)base = n;
    if (!finite) a.status[lr] = WSX_READ_FIT_SMOOTH;
}

// splev (ext = 0) of the fitted cubic at every sample of the read.
__global__ __launch_bounds__(256) void eval_kernel(EvalArgs a)
{
    const int lr = blockIdx.x; // one block per read, striding over its samples
    if (a.status[lr] != 0) return;
    const int r = a.first_read + lr;
    const long long off = a.offsets[r] - a.base_off;
    const int T = (int)(a.offsets[r + 1] - a.offsets[r]);
    if (a.smooth_slot && a.smooth_slot[lr] >= 0) { // a smoothing spline with interior knots: the interval by bisection
        const double *base = a.smooth_ws + (size_t)a.smooth_slot[lr] * smooth_slot_doubles(a.smooth_nest);
        const int n = *(const int *)base, nk1 = n - 4;
        const double *t = base + 8, *c = t + (a.smooth_nest + 2);
        for (int i = threadIdx.x; i < T; i += 256) {
            const double arg = a.signal[off + i];
            // splev's search ends at the largest l in [4, nk1] with t[l] <= arg (4 if there is none)
            int lo = 4, hi = nk1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (t[mid] <= arg) lo = mid;
                else hi = mid - 1;
            }
            double h[6];
            bspl_knots(t, arg, lo, h);
            double sp = 0.0;
#pragma unroll
            for (int j = 1; j <= 4; j++) sp = sp + c[lo - 4 + j] * h[j];
            a.out[off + i] = sp;
            if (a.out_user) a.out_user[off + i] = sp;
        }
        return;
    }
    const double *co = a.coef + (size_t)lr * 6;
    const double xb = co[0], xe = co[1], c1 = co[2], c2 = co[3], c3 = co[4], c4 = co[5];
    const double den = xe - xb, rden = wsx_bspl_rden(den);
    for (int i = threadIdx.x; i < T; i += 256) {
        double h[5];
        bspl4(xb, xe, den, rden, a.signal[off + i], h);
        double sp = 0.0;
        sp = sp + c1 * h[1];
        sp = sp + c2 * h[2];
        sp = sp + c3 * h[3];
        sp = sp + c4 * h[4];
        a.out[off + i] = sp;
        if (a.out_user) a.out_user[off + i] = sp;
    }
}

} // namespace

hipError_t wsx_launch_mid(const MidArgs &a, int max_T, const WsxTuning &tun, hipStream_t s)
{
    if (a.n_reads <= 0) return hipSuccess;
    const int m = a.prm.m;
    // a run spans >= m-1 samples (except possibly the first and last): bound on runs per read
    const int max_runs = max_T / (m - 1 > 0 ? m - 1 : 1) + 2;
    if (a.n_align) hipLaunchKernelGGL(reps_stats_kernel, dim3((a.n_reads + 63) / 64), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(run_stats_kernel, dim3(a.n_reads, std::min((max_runs + 63) / 64, RS_GROUPS)), dim3(64), 0, s, a);
    // small launches: a wavefront per read (latency); big ones: a thread per read (throughput; the wavefront form at 25 000
    // reads per launch: 13.9 instead of 13.4 ms per 100 k-read step, scripts/r03_cycle33.sh)
    const int borders_wave_below = tun.borders_wave_below;
    if (a.n_reads < borders_wave_below) hipLaunchKernelGGL(borders_kernel, dim3((a.n_reads + 3) / 4), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(borders_thread_kernel, dim3((a.n_reads + 63) / 64), dim3(64), 0, s, a);
    if (a.seq_out) hipLaunchKernelGGL(sequence_kernel, dim3((a.n_reads + 3) / 4), dim3(256), 0, s, a);
    if (a.pass == 1) {
        const int max_chunks = max_runs / a.prm.states_in_segment + 2;
        const bool two_kernels = tun.segment_two_kernels != 0; // (tests force the fallback this way)
        const size_t seg_lds = (size_t)(2 * max_chunks + 1) * sizeof(int);
        if (seg_lds <= 40 * 1024 && !two_kernels) {
            hipLaunchKernelGGL(segment_kernel, dim3(a.n_reads), dim3(256), seg_lds, s, a, max_chunks);
        } else { // reads of > ~90k samples: t-statistics through HBM, one thread per chunk
            hipLaunchKernelGGL(tstat_kernel, dim3(a.n_reads, (max_T + 255) / 256), dim3(256), 0, s, a);
            hipLaunchKernelGGL(chunk_kernel, dim3(a.n_reads, (max_chunks + 63) / 64), dim3(64), 0, s, a);
        }
        hipLaunchKernelGGL(sort_kernel, dim3(a.n_reads), dim3(64), 0, s, a);
    }
    return hipGetLastError();
}

hipError_t wsx_launch_fit(const FitArgs &a, hipStream_t s)
{
    if (a.n_reads <= 0) return hipSuccess;
    hipLaunchKernelGGL(fit_kernel, dim3((a.n_reads + 15) / 16), dim3(64), 0, s, a); // a quad of lanes per read
    return hipGetLastError();
}

size_t wsx_smooth_workspace_bytes(int count, int max_m) { return (size_t)count * smooth_slot_doubles(max_m + 4 > 9 ? max_m + 4 : 9) * 8; }

hipError_t wsx_launch_fit_smooth(const FitArgs &a, int count, int max_m, double *ws, hipStream_t s)
{
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(fit_smooth_kernel, dim3((count + 63) / 64), dim3(64), 0, s, a, count, max_m + 4 > 9 ? max_m + 4 : 9, ws);
    return hipGetLastError();
}

hipError_t wsx_launch_eval(const EvalArgs &a, int max_T, hipStream_t s)
{
    if (a.n_reads <= 0 || max_T <= 0) return hipSuccess;
    hipLaunchKernelGGL(eval_kernel, dim3(a.n_reads), dim3(256), 0, s, a);
    return hipGetLastError();
}
