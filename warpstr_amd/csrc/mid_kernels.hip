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
// Parallelisation:
//   mid_kernel   one wavefront per read; lanes stride over the read's runs (alignment records),
//                over its segmentation chunks and over the rank computation of the stable sort.
//   fit_kernel   one THREAD per read: the Givens triangularisation is a sequential recurrence over
//                the sorted points (4 rotations, each 3 divisions + 1 sqrt, per point), so reads are
//                the parallel axis.
//   eval_kernel  one thread per sample: cubic B-spline evaluation (de Boor recurrence, 6 divisions).
#include "../../include/warpstr_hip.h"
#include "wsx_device.h"

namespace {

constexpr double kInf = __builtin_huge_val();

__device__ __forceinline__ int rfl(int x) { return __builtin_amdgcn_readfirstlane(x); }

__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    long long b = __double_as_longlong(v);
    int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
    int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

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
struct LoadAbsDiff {
    const double *a, *b;
    __device__ double operator()(int i) const { return fabs(a[i] - b[i]); }
};

template <class L>
__device__ double pw_block(const L &ld, int o, int n) // n <= 128
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

// Explicit-stack form of the recursion  sum(a,n) = sum(a,n2) + sum(a+n2,n-n2),  n2 = n/2 - (n/2)%8.
template <class L>
__device__ double np_pairwise_sum(const L &ld, int n)
{
    if (n <= 128) return pw_block(ld, 0, n);
    // post-order evaluation; depth <= 24 covers n up to 2^31
    int so[26], sn[26];
    double sv[26];
    unsigned char st[26];
    int sp = 0;
    so[0] = 0;
    sn[0] = n;
    st[0] = 0;
    double ret = 0.0;
    while (sp >= 0) {
        const int o = so[sp], nn = sn[sp];
        if (nn <= 128) {
            ret = pw_block(ld, o, nn);
            sp--;
            continue;
        }
        int n2 = nn / 2;
        n2 -= n2 % 8;
        if (st[sp] == 0) {
            st[sp] = 1;
            sp++;
            so[sp] = o;
            sn[sp] = n2;
            st[sp] = 0;
        } else if (st[sp] == 1) {
            sv[sp] = ret; // left result
            st[sp] = 2;
            sp++;
            so[sp] = o + n2;
            sn[sp] = nn - n2;
            st[sp] = 0;
        } else {
            ret = sv[sp] + ret;
            sp--;
        }
    }
    return ret;
}

__device__ double np_mean(const double *a, int n) { return np_pairwise_sum(LoadPlain{a}, n) / (double)n; }

__device__ double np_std(const double *a, int n)
{
    const double mean = np_pairwise_sum(LoadPlain{a}, n) / (double)n;
    return sqrt(np_pairwise_sum(LoadSqDev{a, mean}, n) / (double)n);
}

// k-th order statistic (0-based) of a[0..n) by rank counting; ties broken by index.
__device__ double select_rank(const double *a, int n, int kth)
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

__device__ double np_median(const double *a, int n)
{
    if (n & 1) return select_rank(a, n, n / 2);
    return (select_rank(a, n, n / 2 - 1) + select_rank(a, n, n / 2)) / 2.0;
}

// ---- sliding t-test segmentation (caller.py:347-378) ------------------------------------------
__device__ __forceinline__ double mean3(const double *a) { return (((0.0 + a[0]) + a[1]) + a[2]) / 3.0; }
__device__ __forceinline__ double std3(const double *a)
{
    const double mu = mean3(a);
    const double d0 = a[0] - mu, d1 = a[1] - mu, d2 = a[2] - mu;
    return sqrt((((0.0 + d0 * d0) + d1 * d1) + d2 * d2) / 3.0);
}

// number of detected events minus one in data[0..n), win = 3; n >= 6 (checked by the caller)
__device__ int segment_count(const double *data, int n)
{
    const int win = 3;
    const int nt = n - 2 * win + 1;
    int borders = 0;
    bool start = false;
    double prev = 0.0;
    for (int q = 0; q < nt; q++) {
        const double *a1 = data + q, *a2 = data + q + win;
        const double s1 = std3(a1), s2 = std3(a2);
        double sd = sqrt((s1 * s1 + s2 * s2) / (double)win);
        if (sd == 0.0) sd = sd + 0.0000001;
        const double t = (mean3(a1) - mean3(a2)) / sd;
        if (q == 0) prev = t;
        if (t > 3 || t < -3) {
            if ((t > 3 && t >= prev) || (t < -3 && t <= prev)) {
                start = true;
            } else {
                if (start) borders++;
                start = false;
            }
        } else if (start) {
            borders++;
            start = false;
        }
        prev = t;
    }
    return borders - 1;
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
__global__ __launch_bounds__(64) void mid_kernel(MidArgs a)
{
    const int lane = threadIdx.x & 63;
    const int lr = blockIdx.x;
    if (lr >= a.n_reads) return;
    const int r = a.first_read + lr;
    wsx_result *res = (wsx_result *)a.results + lr;
    int status = a.status[lr];
    if (status != 0) {
        if (lane == 0 && a.pass == 1) {
            res->status = status;
            res->len1 = res->len2 = res->n_trans1 = res->n_trans2 = 0;
            res->reserved = 0;
            res->cost1 = res->cost2 = __builtin_nan("");
            res->dtw_end_cost1 = res->dtw_end_cost2 = kInf;
        } else if (lane == 0) {
            res->status = status;
        }
        return;
    }
    const long long off = a.offsets[r] - a.base_off;
    const int T = (int)(a.offsets[r + 1] - a.offsets[r]);
    const DevAutomaton A = a.aut[a.aut_id[r]];
    const double *sig = a.signal + off;
    const int n = a.n_runs[lr];
    const uint16_t *rs = a.run_state + off; // reverse time order
    const int32_t *rst = a.run_start + off;
    auto fstate = [&](int k) -> int { return rs[n - 1 - k]; };
    auto fstart = [&](int k) -> int { return rst[n - 1 - k]; };
    auto fend = [&](int k) -> int { return (k == n - 1) ? T : rst[n - 2 - k]; }; // exclusive
    const int m = a.prm.m;
    const int sis = a.prm.states_in_segment;

    // ---- (1) alignment records: one per run (create_alignment, reps_as_one = False) ------------
    double *alv = a.al_value + off, *ale = a.al_expected + off, *alc = a.al_cost + off;
    uint8_t *alg = a.al_good + off;
    int n_good_local = 0;
    for (int k = lane; k < n; k += 64) {
        const int s0 = fstart(k), len = fend(k) - s0;
        const int st = fstate(k);
        const double *raw = sig + s0;
        const double val = a.prm.method_median ? np_median(raw, len) : np_mean(raw, len);
        const double expd = A.value[st];
        const bool good = (len >= m) && (np_std(raw, len) < a.prm.max_std) && (fabs(expd - val) <= a.prm.threshold);
        alv[k] = val;
        ale[k] = expd;
        alc[k] = fabs(val - expd);
        alg[k] = good ? 1 : 0;
        n_good_local += good ? 1 : 0;
    }
    // total number of good records
    int n_good = n_good_local;
    for (int o = 32; o > 0; o >>= 1) n_good += __shfl_xor(n_good, o);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();

    // ---- (2) sequence length after flank stripping (_get_sequence) -----------------------------
    int slo, shi;
    py_slice((long long)A.flank_length - A.seq_idx[fstate(0)], -(long long)A.flank_length, n, &slo, &shi);
    const int seqlen = shi - slo;

    // ---- (3) find_event_borders ----------------------------------------------------------------
    int start = -1, end = -1;
    for (int c = 0; c * 64 < n; c++) {
        const int k = c * 64 + lane;
        const bool isrep = (k < n) && A.repeat_mask[fstate(k)];
        const unsigned long long bal = __ballot(isrep);
        if (bal) {
            if (start < 0) start = c * 64 + __builtin_ctzll(bal);
            end = c * 64 + 63 - __builtin_clzll(bal);
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
                const int es = fstate(end);
                int eR = -1;
                for (int c = 0; c * 64 < n; c++) {
                    const int k = c * 64 + lane;
                    const unsigned long long bal = __ballot((k < n) && fstate(k) == es);
                    if (bal) eR = c * 64 + 63 - __builtin_clzll(bal);
                }
                nb = eR - start;
            }
        }
        if (status == 0) {
            nsel = nb > 0 ? (nb + sis - 1) / sis : 0;
            if (nsel == 0) status = WSX_READ_SEGMENT_RANGE;
        }
    }
    // sel(q) = last sample of run start + q*sis
    auto sel = [&](int q) -> int { return fend(start + q * sis) - 1; };

    // ---- (4) chunk checks (both passes) and segmentation + mask (pass 1) -----------------------
    uint32_t *mw = a.maskbits ? a.maskbits + (off / 32 + lr) : nullptr;
    const int nwords = (T + 31) / 32;
    if (a.pass == 1 && mw) {
        for (int w = lane; w < nwords; w += 64) mw[w] = 0;
        if (a.badmask_bytes)
            for (int q = lane; q < T; q += 64) a.badmask_bytes[off + q] = 0;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    if (status == 0) {
        bool bad_range = false;
        for (int c = lane; c < nsel - 1; c += 64) {
            const int b0 = sel(c), b1 = sel(c + 1);
            const int lo = b0 - 3;
            int hi = b1 + 3;
            if (hi > T) hi = T;
            if (lo < 0 || hi - lo - 5 <= 0) {
                bad_range = true;
                continue;
            }
            if (a.pass == 1) {
                const int cnt = segment_count(sig + lo, hi - lo);
                if (cnt >= sis + 1 && mw) {
                    for (int q = b0; q < b1; q++) {
                        atomicOr(&mw[q >> 5], 1u << (q & 31));
                        if (a.badmask_bytes) a.badmask_bytes[off + q] = 1;
                    }
                }
            }
        }
        if (__ballot(bad_range)) status = WSX_READ_SEGMENT_RANGE;
    }

    // ---- (5) state-wise cost over alignment[start:end] ------------------------------------------
    double cost = __builtin_nan("");
    if (status == 0) {
        int clo, chi;
        py_slice(start, end, n, &clo, &chi);
        if (chi > clo) cost = np_pairwise_sum(LoadPlain{alc + clo}, chi - clo) / (double)(chi - clo);
    }

    // ---- (6) rescaling input: good records, stably sorted by value ------------------------------
    if (status == 0 && a.pass == 1) {
        double *fx = a.fit_x + off, *fy = a.fit_y + off;
        for (int kb = 0; kb < n; kb += 64) {
            const int k = kb + lane;
            const bool mine = (k < n) && alg[k];
            const double xk = (k < n) ? alv[k] : 0.0;
            int rank = 0;
            for (int qb = 0; qb < n; qb += 64) {
                const int qq = qb + lane;
                const double xq_l = (qq < n) ? alv[qq] : 0.0;
                const unsigned long long gq = __ballot((qq < n) && alg[qq]);
                const int lim = (n - qb) < 64 ? (n - qb) : 64;
                for (int t = 0; t < lim; t++) {
                    if (!((gq >> t) & 1ull)) continue;
                    const double xq = readlane_f64(xq_l, t);
                    const int q = qb + t;
                    rank += (xq < xk) || (xq == xk && q < k);
                }
            }
            if (mine) {
                fx[rank] = xk;
                fy[rank] = ale[k];
            }
        }
        if (lane == 0) a.fit_m[lr] = n_good;
    }

    // ---- (7) results ----------------------------------------------------------------------------
    if (lane == 0) {
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
        } else {
            res->len2 = seqlen;
            res->n_trans2 = n;
            res->cost2 = cost;
        }
        if (a.end_cost) {
            if (a.pass == 1) {
                res->dtw_end_cost1 = a.end_cost[lr];
                res->dtw_end_cost2 = kInf;
            } else {
                res->dtw_end_cost2 = a.end_cost[lr];
            }
        }
    }
}

// ---- FITPACK pieces ---------------------------------------------------------------------------
// fpbspl for the 8-knot cubic (knots xb x4, xe x4; interval l = 4): the 4 non-zero B-splines at x.
__device__ __forceinline__ void bspl4(double xb, double xe, double x, double h[5])
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
            const double f = hh[i] / (xe - xb);
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

// fpcurf (iopt = 0, k = 3, s = m, unit weights) restricted to its first iteration, + fpback.
__global__ __launch_bounds__(64) void fit_kernel(FitArgs a)
{
    const int lr = blockIdx.x * blockDim.x + threadIdx.x;
    if (lr >= a.n_reads) return;
    if (a.status[lr] != 0) return;
    const int r = a.first_read + lr;
    const long long off = a.offsets[r] - a.base_off;
    const int m = a.fit_m[lr];
    const double *x = a.fit_x + off, *y = a.fit_y + off;
    const double xb = x[0], xe = x[m - 1];
    if (!(xb < xe)) {
        a.status[lr] = WSX_READ_FIT_ORDER;
        return;
    }
    // banded upper-triangular A (4x4, row j: a[j][1..4]) and right-hand side z
    double A11 = 0, A12 = 0, A13 = 0, A14 = 0, A21 = 0, A22 = 0, A23 = 0, A31 = 0, A32 = 0, A41 = 0;
    double z1 = 0, z2 = 0, z3 = 0, z4 = 0, fp = 0.0;
    for (int it = 0; it < m; it++) {
        double h[5];
        bspl4(xb, xe, x[it], h);
        double yi = y[it] * 1.0;
        h[1] = h[1] * 1.0;
        h[2] = h[2] * 1.0;
        h[3] = h[3] * 1.0;
        h[4] = h[4] * 1.0;
        double c, s;
        // i = 1 -> row 1
        if (h[1] != 0.0) {
            givens(h[1], A11, c, s);
            rota(c, s, yi, z1);
            rota(c, s, h[2], A12);
            rota(c, s, h[3], A13);
            rota(c, s, h[4], A14);
        }
        if (h[2] != 0.0) {
            givens(h[2], A21, c, s);
            rota(c, s, yi, z2);
            rota(c, s, h[3], A22);
            rota(c, s, h[4], A23);
        }
        if (h[3] != 0.0) {
            givens(h[3], A31, c, s);
            rota(c, s, yi, z3);
            rota(c, s, h[4], A32);
        }
        if (h[4] != 0.0) {
            givens(h[4], A41, c, s);
            rota(c, s, yi, z4);
        }
        fp = fp + yi * yi;
    }
    if (!(fp < (double)m)) {
        a.status[lr] = WSX_READ_FIT_SMOOTH;
        return;
    }
    // fpback (n = 4, bandwidth 4)
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
    double *co = a.coef + (size_t)lr * 6;
    co[0] = xb;
    co[1] = xe;
    co[2] = c1;
    co[3] = c2;
    co[4] = c3;
    co[5] = c4;
}

// splev (ext = 0) of the fitted cubic at every sample of the read.
__global__ __launch_bounds__(256) void eval_kernel(EvalArgs a)
{
    const int lr = blockIdx.x; // reads on grid.x (no 65535 limit), 256-sample tiles on grid.y
    if (a.status[lr] != 0) return;
    const int r = a.first_read + lr;
    const long long off = a.offsets[r] - a.base_off;
    const int T = (int)(a.offsets[r + 1] - a.offsets[r]);
    const int i = blockIdx.y * blockDim.x + threadIdx.x;
    if (i >= T) return;
    const double *co = a.coef + (size_t)lr * 6;
    const double xb = co[0], xe = co[1];
    double h[5];
    bspl4(xb, xe, a.signal[off + i], h);
    double sp = 0.0;
    sp = sp + co[2] * h[1];
    sp = sp + co[3] * h[2];
    sp = sp + co[4] * h[3];
    sp = sp + co[5] * h[4];
    a.out[off + i] = sp;
    if (a.out_user) a.out_user[off + i] = sp;
}

} // namespace

hipError_t wsx_launch_mid(const MidArgs &a, hipStream_t s)
{
    if (a.n_reads <= 0) return hipSuccess;
    hipLaunchKernelGGL(mid_kernel, dim3(a.n_reads), dim3(64), 0, s, a);
    return hipGetLastError();
}

hipError_t wsx_launch_fit(const FitArgs &a, hipStream_t s)
{
    if (a.n_reads <= 0) return hipSuccess;
    hipLaunchKernelGGL(fit_kernel, dim3((a.n_reads + 63) / 64), dim3(64), 0, s, a);
    return hipGetLastError();
}

hipError_t wsx_launch_eval(const EvalArgs &a, int max_T, hipStream_t s)
{
    if (a.n_reads <= 0 || max_T <= 0) return hipSuccess;
    hipLaunchKernelGGL(eval_kernel, dim3(a.n_reads, (max_T + 255) / 256), dim3(256), 0, s, a);
    return hipGetLastError();
}
