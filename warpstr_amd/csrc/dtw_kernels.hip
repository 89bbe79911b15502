// dtw_kernels.hip -- DTW over a k-mer state automaton: DP fill + traceback, one read per wavefront.
//
// What it computes is WarpSTR.warp (upstream src/caller/caller.py:189-193):
//   _calc_dtw_astates (198-245)  D[i,j] = min( D[i-1,j] + |s_i - v_j|,                                   "stay"
//                                              min_p ((..(D[i-back,p] + |s_{i-back+1} - v_p|) + ..)
//                                                       + |s_{i-1} - v_p|) + |s_i - v_j| )               "enter from p"
//                                back = m-1 on masked samples else m; strict '<', stay first, then `incoming` order
//   _backtracking (247-301)      path from (T-1, endstate) to row 0
// in fp64 with the reference's left-to-right order of additions, so that D is bit-identical and the
// path is identical.  The traceback uses stored arg-min pointers (0 = stay, f+1 = f-th predecessor),
// which is equivalent to the reference's "closest re-computed candidate" rule (stay wins ties,
// first predecessor wins ties) because the re-computation is exact.
//
// Mapping (register-resident kernel `dtw_pass_fast<M,K,F,MASKED>`):
//   one 64-lane wavefront per read; state j lives in lane j%64, slot j/64 (K slots per lane);
//   a row (= one signal sample) is processed per step: all states of a row are independent.
//   Per state the "dwell" partial sums are kept as a shift register g[1..M-1] that runs one row
//   AHEAD of the DP:   after row i   g[s] = D[i-s+1,j] + |s_{i-s+2}-v_j| + .. + |s_{i+1}-v_j|
//   so g[1] is the next row's stay candidate, and the value a successor needs at row i+2 is already
//   final at the end of row i:  E_j(i+2) = g[M-1] (unmasked row)  or  g[M-2] (masked row, M >= 3).
//   E values are exchanged through LDS (one 8-byte slot per state, double buffered by row parity);
//   a consumer's LDS reads for row i+1 are issued during row i, a full row before they are needed.
//   Absent predecessors point at a slot that holds +inf.  Back-pointers are packed PB bits per row
//   per state into 32-bit words (R rows per word) and written coalesced to a per-read HBM scratch
//   (256 B per wave-store); the traceback scans them word-wise (count-leading-zeros to jump over
//   runs of "stay") and emits run-length state lists plus, optionally, the per-sample trace.
//
// Roofline: the recurrence is a min-plus scan; per row and state ~ (3 + 4F) fp64 VALU ops.  HBM
// traffic per read and pass: 8T (signal) + T*S*PB/8 (pointer scratch, written once, read sparsely)
// + 2T (trace) -- far below what HBM can deliver; the kernel is bound by fp64 VALU issue
// (wave64 fp64 op = 4 cycles on a SIMD).
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

// python-style ceil for positive ints
__device__ __forceinline__ int cdiv(int a, int b) { return (a + b - 1) / b; }

// ------------------------------------------------------------------------------------------------
// Traceback shared by both DP kernels.  Executed redundantly by every lane of the wave (all values
// are wave-uniform); lanes only diverge when storing the per-sample trace.
//   bp words: word (wi, j) at bp[(wi*K + j/64)*64 + j%64], PB bits per row, R = 32/PB rows per word.
// Runs are appended in reverse time order: run_state[q], run_start[q]; adjacent equal states merge
// (self-loop states), matching the run-length encoding of the trace (caller.py:58-60).
// ------------------------------------------------------------------------------------------------
template <int PB>
__device__ void traceback(const DevAutomaton &A, const uint32_t *bp, int K, int T, int m,
                          const uint32_t *maskw /* packed mask of this read or NULL */, uint16_t *run_state,
                          int32_t *run_start, int32_t *n_runs_out, uint16_t *trace, int lane)
{
    constexpr int R = 32 / PB;
    constexpr uint32_t PM = (1u << PB) - 1u;
    int j = A.endstate;
    int i = T - 1;
    int run_end = T - 1;
    int nr = 0;
    int last_state = -1;
    while (true) {
        const int k = j >> 6, ln = j & 63;
        int wi = i / R;
        uint32_t w = bp[((size_t)wi * K + k) * 64 + ln];
        const int sh = (i % R + 1) * PB;
        uint32_t wm = (sh >= 32) ? w : (w & ((1u << sh) - 1u));
        while (wm == 0 && wi > 0) {
            wi--;
            wm = bp[((size_t)wi * K + k) * 64 + ln];
        }
        wm = (uint32_t)rfl((int)wm);
        wi = rfl(wi);
        int start, ptr = 0;
        if (wm == 0) {
            start = 0;
        } else {
            const int top = 31 - __builtin_clz(wm);
            const int rr = top / PB;
            start = wi * R + rr;
            ptr = (int)((wm >> (rr * PB)) & PM);
        }
        // emit run [start, run_end] of state j
        if (trace) {
            for (int q = start + lane; q <= run_end; q += 64) trace[q] = (uint16_t)j;
        }
        if (j == last_state) {
            if (lane == 0) run_start[nr - 1] = start;
        } else {
            if (lane == 0) {
                run_state[nr] = (uint16_t)j;
                run_start[nr] = start;
            }
            nr++;
            last_state = j;
        }
        if (wm == 0) break;
        const int p = A.pred_idx[A.pred_ptr[j] + ptr - 1];
        int back = m;
        if (maskw) back = ((maskw[start >> 5] >> (start & 31)) & 1u) ? m - 1 : m;
        run_end = start - 1;
        i = start - back;
        j = rfl(p);
        if (i < 0) break; // cannot happen: pointers are only set on rows >= m
    }
    if (lane == 0) *n_runs_out = nr;
}

// ------------------------------------------------------------------------------------------------
// Register-resident DP (see file header).
// ------------------------------------------------------------------------------------------------
template <int M, int K, int F, bool MASKED>
__global__ __launch_bounds__(256) void dtw_pass_fast(PassArgs a)
{
    static_assert(M == 4, "register-resident kernel is specialised for min_values_per_state = 4");
    constexpr int PB = (F <= 3) ? 2 : 4;
    constexpr int R = 32 / PB;
    constexpr int EXW = K * 64 + 1; // export slots per buffer (+1: the +inf slot)
    extern __shared__ double lds[];

    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int slot = rfl(blockIdx.x * 4 + wib);
    if (slot >= a.n_launch) return;
    const int r = rfl(a.order[slot]);
    const int lr = r - a.first_read;
    const long long off = a.offsets[r] - a.base_off;
    const int T = (int)(a.offsets[r + 1] - a.offsets[r]);
    const DevAutomaton A = a.aut[a.aut_id[r]];
    const int S = A.n_states;
    if (a.check_status && a.status[lr] != 0) return;
    if (T <= M || S <= M) {
        if (lane == 0) {
            a.status[lr] = 1; // WSX_READ_SHAPE
            a.n_runs[lr] = 0;
            if (a.end_cost) a.end_cost[lr] = kInf;
        }
        return;
    }
    const double *sig = a.signal + off;
    double *ex = lds + wib * (2 * EXW);
    uint32_t *bp = a.bp + (size_t)(off / R + lr) * (K * 64);
    const uint32_t *maskw = MASKED ? (a.maskbits + (off / 32 + lr)) : nullptr;

    // ---- per-state constants -----------------------------------------------------------------
    double v[K];
    int paddr[K][F];
    bool cutf[K];
    const long long boundary = (long long)A.flank_length - 10;
    const long long after_repeat = (long long)A.seq_idx_last - boundary;
    long long cut_from_ll = 6 * boundary; // rows i >= first_threshold and i > second_threshold
    if ((long long)T - 6 * boundary + 1 > cut_from_ll) cut_from_ll = (long long)T - 6 * boundary + 1;
    const int cut_from = cut_from_ll < 0 ? 0 : (cut_from_ll > T ? T : (int)cut_from_ll);
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int j = k * 64 + lane;
        const bool valid = j < S;
        v[k] = valid ? A.value[j] : 0.0;
        cutf[k] = valid && ((long long)A.seq_idx[j] < after_repeat);
        const int pp = valid ? A.pred_ptr[j] : 0;
        const int nf = valid ? (A.pred_ptr[j + 1] - pp) : 0;
#pragma unroll
        for (int f = 0; f < F; f++) paddr[k][f] = (f < nf) ? A.pred_idx[pp + f] : (EXW - 1);
    }
    if (lane == 0) {
        ex[EXW - 1] = kInf;
        ex[2 * EXW - 1] = kInf;
    }

    // ---- row 0 (caller.py:201-208) -----------------------------------------------------------
    const double v0 = A.value[0];
    const double start_val = fabs(sig[0] - v0);
    double g1[K], g2[K], g3[K]; // the ahead pipeline (named registers: no dynamic indexing)
    double d[K];
    double acur[K];
    uint32_t bpw[K];
    const double s1 = sig[1];
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int j = k * 64 + lane;
        double d0 = kInf;
        if (j == 0) d0 = start_val;
        else if (j <= M && j < S) d0 = start_val + fabs(sig[j] - v0);
        d[k] = d0;
        acur[k] = fabs(s1 - v[k]);
        g1[k] = d0 + acur[k];
        g2[k] = kInf;
        g3[k] = kInf;
        bpw[k] = 0;
        ex[0 * EXW + j] = kInf;        // E(2): never used (rows < M are forced to inf) but defined
        ex[1 * EXW + j] = kInf;        // E(1)
    }
    __builtin_amdgcn_wave_barrier();
    double ecur[K][F];
#pragma unroll
    for (int k = 0; k < K; k++)
#pragma unroll
        for (int f = 0; f < F; f++) ecur[k][f] = ex[1 * EXW + paddr[k][f]];

    // signal samples are fetched 64 at a time (one coalesced 512-B load) and broadcast by readlane
    int blk = 0; // cur holds samples [blk*64, blk*64+64)
    auto clampi = [&](int x) { return x < T ? x : T - 1; };
    double cur = sig[clampi(lane)];
    double nxt = sig[clampi(64 + lane)];
    uint32_t mwords = 0; // packed mask words [mblk*64 .. +64) of this read, one per lane
    int mblk = 0;
    if (MASKED) mwords = maskw[lane < cdiv(T, 32) ? lane : 0];

    for (int i = 1; i < T; i++) {
        // s_{i+1}
        const int idx = clampi(i + 1);
        if ((idx >> 6) != blk) {
            blk = idx >> 6;
            cur = nxt;
            nxt = sig[clampi((blk + 1) * 64 + lane)];
        }
        const double snext = readlane_f64(cur, idx & 63);
        bool mask_i2 = false;
        if (MASKED) {
            const int i2 = i + 2;
            if (i2 < T) {
                const int wq = i2 >> 5;
                if ((wq >> 6) != mblk) {
                    mblk = wq >> 6;
                    const int widx = mblk * 64 + lane;
                    mwords = maskw[widx < cdiv(T, 32) ? widx : 0];
                }
                const uint32_t mw = (uint32_t)__builtin_amdgcn_readlane((int)mwords, wq & 63);
                mask_i2 = (mw >> (i2 & 31)) & 1u;
            }
        }
        const bool forced = i < M;
        const bool cut_row = i >= cut_from;
        const int wbuf = (i & 1) * EXW;        // E(i+2) goes here
        const int rbuf = ((i + 1) & 1) * EXW;  // E(i+1) was written at the end of row i-1
        const int shift = (i % R) * PB;
#pragma unroll
        for (int k = 0; k < K; k++) {
            double best = g1[k];
            uint32_t ptr = 0;
#pragma unroll
            for (int f = 0; f < F; f++) {
                const double cand = ecur[k][f] + acur[k];
                if (cand < best) {
                    best = cand;
                    ptr = f + 1;
                }
            }
            if (forced || (cut_row && cutf[k])) {
                best = kInf;
                ptr = 0;
            }
            const double an = fabs(snext - v[k]);
            const double n3 = g2[k] + an, n2 = g1[k] + an;
            g3[k] = n3;
            g2[k] = n2;
            g1[k] = best + an;
            d[k] = best;
            acur[k] = an;
            double e_out = n3;
            if (MASKED) e_out = mask_i2 ? n2 : n3;
            ex[wbuf + k * 64 + lane] = e_out;
            bpw[k] |= ptr << shift;
        }
        if ((i % R) == R - 1 || i == T - 1) {
            const int wi = i / R;
#pragma unroll
            for (int k = 0; k < K; k++) {
                bp[((size_t)wi * K + k) * 64 + lane] = bpw[k];
                bpw[k] = 0;
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < K; k++)
#pragma unroll
            for (int f = 0; f < F; f++) ecur[k][f] = ex[rbuf + paddr[k][f]];
    }

    // ---- outputs of the fill -----------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int j = k * 64 + lane;
        if (j == A.endstate && a.end_cost) a.end_cost[lr] = d[k];
        if (a.last_row && j < S) a.last_row[(size_t)lr * a.last_row_stride + j] = d[k];
    }
    if (lane == 0 && !a.check_status) a.status[lr] = 0;
    // make the pointer words visible to every lane of this wave before the traceback reads them
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    traceback<PB>(A, bp, K, T, M, maskw, a.run_state + off, a.run_start + off, a.n_runs + lr,
                  a.trace ? a.trace + off : nullptr, lane);
}

// ------------------------------------------------------------------------------------------------
// General DP: any m >= 2, any fan-in <= 15, any S that fits the LDS ring.  A direct data-parallel
// statement of caller.py:217-244: the last m+1 rows of D live in an LDS ring, states are strided over
// the lanes, dwell sums are recomputed per candidate.  Slow path for unusual configurations.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void dtw_pass_generic(PassArgs a, int K)
{
    constexpr int PB = 4;
    constexpr int R = 32 / PB;
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63;
    const int slot = blockIdx.x;
    if (slot >= a.n_launch) return;
    const int r = rfl(a.order[slot]);
    const int lr = r - a.first_read;
    const long long off = a.offsets[r] - a.base_off;
    const int T = (int)(a.offsets[r + 1] - a.offsets[r]);
    const DevAutomaton A = a.aut[a.aut_id[r]];
    const int S = A.n_states;
    const int m = a.m;
    if (a.check_status && a.status[lr] != 0) return;
    if (T <= m || S <= m) {
        if (lane == 0) {
            a.status[lr] = 1;
            a.n_runs[lr] = 0;
            if (a.end_cost) a.end_cost[lr] = kInf;
        }
        return;
    }
    const double *sig = a.signal + off;
    uint32_t *bp = a.bp + (size_t)(off / R + lr) * (K * 64);
    const uint32_t *maskw = a.maskbits ? (a.maskbits + (off / 32 + lr)) : nullptr;
    const int ring = m + 1;
    const int SP = K * 64;
    // ring row q holds D[i, :] for i % ring == q
    for (int q = lane; q < ring * SP; q += 64) lds[q] = kInf;
    __builtin_amdgcn_wave_barrier();
    const double v0 = A.value[0];
    const double start_val = fabs(sig[0] - v0);
    if (lane == 0) lds[0] = start_val;
    if (lane >= 1 && lane <= m && lane < S) lds[lane] = start_val + fabs(sig[lane] - v0);
    // rows 1..m-1 stay inf
    const long long boundary = (long long)A.flank_length - 10;
    const long long after_repeat = (long long)A.seq_idx_last - boundary;
    const long long first_threshold = 6 * boundary, second_threshold = (long long)T - 6 * boundary;
    for (int k = 0; k < K; k++) bp[(size_t)k * 64 + lane] = 0; // word 0 default
    __builtin_amdgcn_wave_barrier();
    uint32_t bpw[WSX_MAX_K * 2];
    for (int k = 0; k < K; k++) bpw[k] = 0;
    for (int i = 1; i < T; i++) {
        const bool real = i >= m;
        int back = m;
        if (maskw && real) back = ((maskw[i >> 5] >> (i & 31)) & 1u) ? m - 1 : m;
        double *row = lds + (i % ring) * SP;
        const double *prow = lds + ((i - 1) % ring) * SP;
        const double *brow = lds + ((i - back + ring) % ring) * SP;
        const double val = sig[i];
        for (int k = 0; k < K; k++) {
            const int j = k * 64 + lane;
            double best = kInf;
            uint32_t ptr = 0;
            if (real && j < S) {
                const long long sj = A.seq_idx[j];
                bool skip = false;
                if (i < first_threshold) skip = (i < sj * 4 && i > sj * 15);
                else if (i > second_threshold && sj < after_repeat) skip = true;
                if (!skip) {
                    const double vj = A.value[j];
                    const double aj = fabs(val - vj);
                    best = prow[j] + aj;
                    const int pp = A.pred_ptr[j], pe = A.pred_ptr[j + 1];
                    for (int e = pp; e < pe; e++) {
                        const int p = A.pred_idx[e];
                        const double vp = A.value[p];
                        double c = brow[p];
                        for (int q = i - back + 1; q < i; q++) c += fabs(sig[q] - vp);
                        c += aj;
                        if (c < best) {
                            best = c;
                            ptr = (uint32_t)(e - pp + 1);
                        }
                    }
                    if (!(best < kInf)) ptr = 0;
                }
            }
            bpw[k] |= ptr << ((i % R) * PB);
            // all reads of this row's inputs (rows i-1 and i-back) are to other ring rows
            if (j < SP) row[j] = best;
        }
        if ((i % R) == R - 1 || i == T - 1) {
            const int wi = i / R;
            for (int k = 0; k < K; k++) {
                bp[((size_t)wi * K + k) * 64 + lane] = bpw[k];
                bpw[k] = 0;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    const double *lrow = lds + ((T - 1) % ring) * SP;
    if (lane == 0) {
        if (a.end_cost) a.end_cost[lr] = lrow[A.endstate];
        if (!a.check_status) a.status[lr] = 0;
    }
    if (a.last_row)
        for (int j = lane; j < S; j += 64) a.last_row[(size_t)lr * a.last_row_stride + j] = lrow[j];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    traceback<PB>(A, bp, K, T, m, maskw, a.run_state + off, a.run_start + off, a.n_runs + lr,
                  a.trace ? a.trace + off : nullptr, lane);
}

template <int M, int K, int F>
hipError_t launch_fast(const PassArgs &a, bool masked, hipStream_t s)
{
    const int blocks = (a.n_launch + 3) / 4;
    const size_t shmem = 4 * 2 * (K * 64 + 1) * sizeof(double);
    if (masked) hipLaunchKernelGGL((dtw_pass_fast<M, K, F, true>), dim3(blocks), dim3(256), shmem, s, a);
    else hipLaunchKernelGGL((dtw_pass_fast<M, K, F, false>), dim3(blocks), dim3(256), shmem, s, a);
    return hipGetLastError();
}

template <int M, int K>
hipError_t launch_fast_f(const PassArgs &a, int F, bool masked, hipStream_t s)
{
    switch (F) {
    case 2: return launch_fast<M, K, 2>(a, masked, s);
    case 3: return launch_fast<M, K, 3>(a, masked, s);
    case 4: return launch_fast<M, K, 4>(a, masked, s);
    }
    return hipErrorInvalidValue;
}

} // namespace

bool wsx_fast_pass_supported(int m, int K, int F) { return m == 4 && K >= 1 && K <= WSX_MAX_K && F >= 1 && F <= WSX_MAX_F; }

static int fast_f(int F) { return F <= 2 ? 2 : F; }

const char *wsx_pass_kernel_name(int m, int K, int F, bool masked, bool generic)
{
    static thread_local char buf[64];
    if (generic) snprintf(buf, sizeof(buf), "dtw_pass_generic");
    else snprintf(buf, sizeof(buf), "dtw_pass_fast<%d,%d,%d,%d>", m, K, fast_f(F), masked ? 1 : 0);
    return buf;
}

hipError_t wsx_launch_pass(const PassArgs &a, int m, int K, int F, bool masked, bool generic, hipStream_t s)
{
    if (a.n_launch <= 0) return hipSuccess;
    if (generic) {
        const size_t shmem = (size_t)(m + 1) * K * 64 * sizeof(double);
        hipLaunchKernelGGL(dtw_pass_generic, dim3(a.n_launch), dim3(64), shmem, s, a, K);
        return hipGetLastError();
    }
    if (m != 4) return hipErrorInvalidValue;
    const int f = fast_f(F);
    switch (K) {
    case 1: return launch_fast_f<4, 1>(a, f, masked, s);
    case 2: return launch_fast_f<4, 2>(a, f, masked, s);
    case 3: return launch_fast_f<4, 3>(a, f, masked, s);
    case 4: return launch_fast_f<4, 4>(a, f, masked, s);
    case 5: return launch_fast_f<4, 5>(a, f, masked, s);
    }
    return hipErrorInvalidValue;
}
