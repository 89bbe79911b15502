// dtw_kernels.hip -- DTW over a k-mer state automaton: DP fill, traceback, trace expansion.
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
// Kernels
//   dtw_fill_fast<M,K,F>        register-resident fill for min_values_per_state M in {3,4,5}: one 64-lane wavefront per
//       read; state j lives in lane j%64, slot j/64 (K slots per lane); one row (= one signal sample) per
//       step, all states of a row are independent.  Per state the "dwell" partial sums are a shift register
//       g_1..g_{M-1} that runs one row AHEAD of the DP:
//             after row i   g_s = D[i-s+1,j] + |s_{i-s+2}-v_j| + .. + |s_{i+1}-v_j|
//       so g_1 is the next row's stay candidate, and the value a successor needs at row i+2 is final at the end
//       of row i:  E_j(i+2) = g_{M-1} (unmasked row) or g_{M-2} (masked row).  E values are exchanged through LDS (one
//       8-byte slot per state, double buffered by row parity); a consumer's LDS reads for row i+1 are issued
//       during row i, a full row before they are needed.  Absent predecessors point at a slot holding +inf.
//       Back-pointers are packed PB bits per row per state into 32-bit words (R rows per word) and written
//       coalesced (256 B per wave-store) to a per-read HBM scratch.
//   dtw_fill_generic            any m >= 2, fan-in <= 15: last m+1 rows of D in an LDS ring, direct restatement.
//   traceback_kernel<PB>        one THREAD per read (the walk is a dependent pointer chase of ~#transitions
//       steps; reads are the parallel axis): scans a state's pointer words downwards with count-leading-zeros
//       to jump over runs of "stay", emits the run-length state list in reverse time order.
//   expand_trace_kernel         optional: per-sample state ids from the run list (coalesced, wave per read).
//
// Roofline: min-plus recurrence, no MFMA.  Per row and state (F = 2): 6 fp64 adds, 2 fp64 compares, 6 selects.
// HBM traffic per read and pass: 8T (signal) + T*K*64*PB/8 (pointer scratch, written once, read sparsely by the
// traceback) + 6*runs; the fill is bound by fp64 VALU issue (a wave64 fp64 op occupies a SIMD for 4 cycles).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

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

__device__ __forceinline__ int cdiv(int a, int b) { return (a + b - 1) / b; }

struct ReadGeom {
    int r, lr, T;
    long long off;
};

__device__ __forceinline__ ReadGeom geom(const PassArgs &a, int slot)
{
    ReadGeom g;
    g.r = a.order[slot];
    g.lr = g.r - a.first_read;
    g.off = a.offsets[g.r] - a.base_off;
    g.T = (int)(a.offsets[g.r + 1] - a.offsets[g.r]);
    return g;
}

// ------------------------------------------------------------------------------------------------
// Register-resident fill (see file header), for min_values_per_state M in {3, 4, 5}.
// ------------------------------------------------------------------------------------------------
template <int M, int K, int F>
struct FillState {
    double v[K], acur[K], d[K]; // acur = s_i - v_j (signed; |.| is a free source modifier)
    double g[K][M];             // g[k][s], s = 1..M-1: the dwell pipeline (g[k][0] unused); always indexed statically
    double e0[K][F], e1[K][F];                       // predecessor exports: row i uses e[i&1], loads e[(i+1)&1]
    int paddr[K][F];                                  // LDS double index of predecessor f's export slot
    uint32_t bpw[K];                                  // pointer bits of the current word, newest row in the low bits
    uint32_t cutclr[K];                               // ~((1<<PB)-1) for corner-cut states, ~0 otherwise
    bool cutf[K];
};

// x + |a| as ONE VALU op: the abs is a source modifier.  (Written as asm because the compiler otherwise
// materialises |a| with two extra 32-bit ops when a is loop-carried.)
__device__ __forceinline__ double add_abs(double x, double a)
{
    double r;
    asm("v_add_f64 %0, %1, |%2|" : "=v"(r) : "v"(x), "v"(a));
    return r;
}

// min(a, b) as one v_min_f64 (the builtin would first canonicalise both operands: two extra VALU ops).
// Operands are sums of finite values or +inf: never NaN.
__device__ __forceinline__ double min_f64(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// bits = (bits << 1) | (cand < best): compare into VCC, then add-with-carry bits+bits+VCC.
__device__ __forceinline__ void push_lt(uint32_t &bits, double cand, double best)
{
    asm("v_cmp_lt_f64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"(cand), "v"(best) : "vcc");
}

// One DP row for all K slots.  PAR = row parity (selects LDS buffers and the e0/e1 roles at compile time),
// FORCED: rows 1..3 (D stays inf, only the pipeline advances), CUT: corner-cut rows.
//   top of row i : issue the LDS reads of E(i+1) (written at the end of row i-1) -- consumed in row i+1
//   body         : D[i,:] from the exports E(i) read one row earlier
//   end of row i : write E(i+2)
// Back-pointer encoding: F bits per row (bit f set <=> predecessor f beat everything before it in the
// reference's order: stay, pred 0, pred 1, ..); the arg-min is the highest set bit; 0 = stay.
// FL < F ("split"): only slot 0 considers F predecessors per state, the other slots FL (the host places every state
// with more than FL predecessors in slot 0; such states are few: loop entries, IUPAC alternatives).  FL == F: uniform.
template <int M, int K, int F, int FL, bool MROW, int PAR, bool FORCED, bool CUT>
__device__ __forceinline__ void dp_row(FillState<M, K, F> &st, double *ex, int lane, double snext)
{
    constexpr int PB = (F <= 2) ? 2 : 4;
    constexpr int EXW = K * 64 + 1;
    constexpr int wbuf = PAR * EXW;       // E(i+2) goes to the buffer of parity i
    constexpr int rbuf = (1 - PAR) * EXW; // E(i+1) lives in the buffer of parity i+1
#pragma unroll
    for (int k = 0; k < K; k++)
#pragma unroll
        for (int f = 0; f < F; f++) {
            if (k > 0 && f >= FL) continue;
            const double e = ex[rbuf + st.paddr[k][f]];
            if (PAR) st.e0[k][f] = e;
            else st.e1[k][f] = e;
        }
#pragma unroll
    for (int k = 0; k < K; k++) {
        double best = st.g[k][1];
        if (FORCED) {
            best = kInf;
            st.bpw[k] <<= PB;
        } else {
            const int Fk = (k > 0) ? FL : F; // folds to a constant once the slot loop is unrolled
            if (PB > Fk) st.bpw[k] <<= (PB - Fk);
            // candidates in the reference's order; bit f lands at position f of this row's field, so push the
            // highest predecessor first
            double cand[F], run[F + 1];
            run[0] = best;
#pragma unroll
            for (int f = 0; f < F; f++) {
                if (f < Fk) {
                    cand[f] = add_abs(PAR ? st.e1[k][f] : st.e0[k][f], st.acur[k]);
                    run[f + 1] = min_f64(run[f], cand[f]);
                } else {
                    cand[f] = kInf;
                    run[f + 1] = run[f];
                }
            }
#pragma unroll
            for (int f = F - 1; f >= 0; f--)
                if (f < Fk) push_lt(st.bpw[k], cand[f], run[f]);
            best = run[F];
            if (CUT) {
                best = st.cutf[k] ? kInf : best;
                st.bpw[k] &= st.cutclr[k];
            }
        }
        const double an = snext - st.v[k];
#pragma unroll
        for (int s = M - 1; s >= 2; s--) st.g[k][s] = add_abs(st.g[k][s - 1], an);
        st.g[k][1] = add_abs(best, an);
        st.d[k] = best;
        st.acur[k] = an;
        // row i+2 masked (back = M-1): successors need the (M-2)-deep sum, else the (M-1)-deep one
        ex[wbuf + k * 64 + lane] = MROW ? st.g[k][M - 2] : st.g[k][M - 1];
    }
    __builtin_amdgcn_wave_barrier();
}

#ifdef WSX_FILL_MAX_WAVES
#define WSX_FILL_OCC __attribute__((amdgpu_waves_per_eu(1, WSX_FILL_MAX_WAVES)))
#else
#define WSX_FILL_OCC
#endif

template <int M, int K, int F, int FL>
__global__ __launch_bounds__(256) WSX_FILL_OCC void dtw_fill_fast(PassArgs a)
{
    static_assert(FL >= 1 && FL <= F, "slots 1.. consider FL <= F predecessors");
    static_assert(M >= 3, "the one-row-ahead export needs min_values_per_state >= 3");
    constexpr int PB = (F <= 2) ? 2 : 4;
    constexpr int R = 32 / PB;
    constexpr int EXW = K * 64 + 1; // export slots per buffer (+1: the +inf slot)
    constexpr int WLDS = 2 * EXW + 136; // doubles of LDS per wave: two export buffers + two 64-sample signal blocks
                                        // (+8: samples 0..7 of the even block mirrored at 128..135, so that eight
                                        // consecutive samples never wrap)
    extern __shared__ double lds[];

    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int slot = rfl(blockIdx.x * 4 + wib);
    if (slot >= a.n_launch) return;
    ReadGeom gm = geom(a, slot);
    const int lr = rfl(gm.lr), T = rfl(gm.T);
    const long long off = gm.off;
    const DevAutomaton A = a.aut[a.aut_id[gm.r]];
    const int S = A.n_states;
    if (a.check_status && a.status[lr] != 0) return;
    if (T <= M || S <= M) {
        if (lane == 0) {
            a.status[lr] = 1; // WSX_READ_SHAPE
            if (a.end_cost) a.end_cost[lr] = kInf;
        }
        return;
    }
    const double *sig = a.signal + off;
    double *ex = lds + wib * WLDS;
    double *sb = ex + 2 * EXW; // signal blocks: sample q lives at sb[q & 127]
    uint32_t *bp = a.bp + (size_t)(off / R + lr) * (K * 64);
    const uint32_t *maskw = a.maskbits ? (a.maskbits + (off / 32 + lr)) : nullptr;
    const int nmw = cdiv(T, 32);

    // ---- per-state constants -----------------------------------------------------------------
    FillState<M, K, F> st;
    const long long boundary = (long long)A.flank_length - 10;
    const long long after_repeat = (long long)A.seq_idx_last - boundary;
    long long cut_from_ll = 6 * boundary; // rows with i >= first_threshold and i > second_threshold
    if ((long long)T - 6 * boundary + 1 > cut_from_ll) cut_from_ll = (long long)T - 6 * boundary + 1;
    const int cut_from = cut_from_ll < M ? M : (cut_from_ll > T ? T : (int)cut_from_ll);
    int sid[K]; // state handled at position k*64 + lane (-1: none)
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int q = k * 64 + lane;
        int j = q < S ? q : -1;
        if (A.state_at) {
            const int t = A.state_at[q];
            j = t == 0xFFFF ? -1 : t;
        }
        sid[k] = j;
        const bool valid = j >= 0;
        st.v[k] = valid ? A.value[j] : 0.0;
        st.cutf[k] = valid && ((long long)A.seq_idx[j] < after_repeat);
        st.cutclr[k] = st.cutf[k] ? ~((1u << PB) - 1u) : ~0u;
        const int pp = valid ? A.pred_ptr[j] : 0;
        const int nf = valid ? (A.pred_ptr[j + 1] - pp) : 0;
#pragma unroll
        for (int f = 0; f < F; f++) {
            int pa = EXW - 1; // the +inf slot
            if (f < nf) {
                const int p = A.pred_idx[pp + f];
                pa = A.pos ? A.pos[p] : p;
            }
            st.paddr[k][f] = pa;
        }
    }
    if (lane == 0) {
        ex[EXW - 1] = kInf;
        ex[2 * EXW - 1] = kInf;
    }

    // ---- row 0 (caller.py:201-208) -----------------------------------------------------------
    const double v0 = A.value[0];
    const double start_val = fabs(sig[0] - v0);
    const double s1 = sig[1];
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int j = sid[k];
        double d0 = kInf;
        if (j == 0) d0 = start_val;
        else if (j > 0 && j <= M) d0 = start_val + fabs(sig[j] - v0);
        st.d[k] = d0;
        st.acur[k] = s1 - st.v[k];
        st.g[k][1] = d0 + fabs(st.acur[k]);
#pragma unroll
        for (int q = 2; q < M; q++) st.g[k][q] = kInf;
        st.bpw[k] = 0;
        ex[0 * EXW + k * 64 + lane] = kInf; // E(2), E(1): never used (rows < M are forced to inf) but defined
        ex[1 * EXW + k * 64 + lane] = kInf;
#pragma unroll
        for (int f = 0; f < F; f++) {
            st.e0[k][f] = kInf;
            st.e1[k][f] = kInf;
        }
    }

    // Signal: 64 samples per coalesced 512-B load, parked in LDS (two blocks) and broadcast to the wave by a
    // same-address ds_read -- no VALU involved.  Row i consumes s_{i+1}; it is read one row early.
    auto clampi = [&](int x) { return x < T ? x : T - 1; };
    sb[lane] = sig[clampi(lane)];
    if (lane < 8) sb[128 + lane] = sb[lane];
    double nxt = sig[clampi(64 + lane)];
    __builtin_amdgcn_wave_barrier();
    double s_even = 0.0, s_odd = 0.0; // s_q for the even / odd q most recently read
    s_even = sb[2];                   // row 1 consumes s_2
    const int last = T - 1;

    for (int b = 0; b * 64 - 1 <= last; b++) {
        sb[((b + 1) & 1) * 64 + lane] = nxt; // block b+1 (row 64b+62 reads s_{64b+64} ahead)
        if (((b + 1) & 1) == 0 && lane < 8) sb[128 + lane] = nxt;
        nxt = sig[clampi((b + 2) * 64 + lane)];
        __builtin_amdgcn_wave_barrier();
        const int base = b * 64 - 1;
        int lo = base, hi = b * 64 + 63; // rows [lo, hi)
        if (lo < 1) lo = 1;
        if (hi > T) hi = T;

        // bm: bit t <=> row base+t exports for a MASKED row (mask bit of sample base+t+2 = 64b+1+t)
        unsigned long long bm = 0;
        if (maskw) {
            const int w0 = 2 * b;
            const unsigned long long m0 = w0 < nmw ? maskw[w0] : 0u, m1 = w0 + 1 < nmw ? maskw[w0 + 1] : 0u,
                                     m2 = w0 + 2 < nmw ? maskw[w0 + 2] : 0u;
            bm = ((m0 | (m1 << 32)) >> 1) | ((m2 & 1ull) << 63);
            bm = ((unsigned long long)(unsigned)rfl((int)(bm >> 32)) << 32) | (unsigned)rfl((int)bm);
        }

        // one row, everything wave-uniform except the per-lane state
        // s_new = s_{i+2}, prefetched (parity of i) while the row consumes s_{i+1} (other parity)
        auto row = [&](auto par, auto forced, auto cut, auto msk, int i, double s_new) {
            constexpr int PAR = decltype(par)::value;
            constexpr bool FORCED = decltype(forced)::value;
            constexpr bool CUT = decltype(cut)::value;
            constexpr bool MROW = decltype(msk)::value;
            const double snext = PAR ? s_even : s_odd;
            if (PAR) s_odd = s_new;
            else s_even = s_new;
            dp_row<M, K, F, FL, MROW, PAR, FORCED, CUT>(st, ex, lane, snext);
            if ((i % R) == R - 1 || i == last) {
                // word complete (row r of the word sits at bits PB*(R-1-r)); left-align a partial last word
                const int wi = i / R;
                const int fill = (R - 1 - (i % R)) * PB;
#pragma unroll
                for (int k = 0; k < K; k++) {
                    bp[((size_t)wi * K + k) * 64 + lane] = st.bpw[k] << fill;
                    st.bpw[k] = 0;
                }
            }
        };
        // rows [plo, phi) with constant compile-time flags.  Rows come two per iteration (the even/odd register roles
        // then need no copies); for K <= 2, aligned groups of eight rows share ONE LDS address computation for their
        // signal samples (immediate offsets 0..56 from it).
        auto span = [&](auto forced, auto cut, auto msk, int plo, int phi) {
            using P0 = std::integral_constant<int, 0>;
            using P1 = std::integral_constant<int, 1>;
            int i = plo;
            if (i < phi && (i & 1)) {
                row(P1{}, forced, cut, msk, i, sb[(i + 2) & 127]);
                i++;
            }
            if constexpr (K <= 2) {
                for (; i + 1 < phi && (i & 7); i += 2) {
                    row(P0{}, forced, cut, msk, i, sb[(i + 2) & 127]);
                    row(P1{}, forced, cut, msk, i + 1, sb[(i + 3) & 127]);
                }
                for (; i + 8 <= phi; i += 8) {
                    const double *sg = sb + ((i + 2) & 127); // <= 122: sg[0..7] stays inside the mirrored buffer
                    row(P0{}, forced, cut, msk, i, sg[0]);
                    row(P1{}, forced, cut, msk, i + 1, sg[1]);
                    row(P0{}, forced, cut, msk, i + 2, sg[2]);
                    row(P1{}, forced, cut, msk, i + 3, sg[3]);
                    row(P0{}, forced, cut, msk, i + 4, sg[4]);
                    row(P1{}, forced, cut, msk, i + 5, sg[5]);
                    row(P0{}, forced, cut, msk, i + 6, sg[6]);
                    row(P1{}, forced, cut, msk, i + 7, sg[7]);
                }
            }
            for (; i + 1 < phi; i += 2) {
                row(P0{}, forced, cut, msk, i, sb[(i + 2) & 127]);
                row(P1{}, forced, cut, msk, i + 1, sb[(i + 3) & 127]);
            }
            if (i < phi) row(P0{}, forced, cut, msk, i, sb[(i + 2) & 127]);
        };
        // a phase, split into maximal runs of equal mask bit so that the row code is branch-free
        auto phase = [&](auto forced, auto cut, int plo, int phi) {
            int i = plo;
            while (i < phi) {
                const unsigned long long rest = bm >> (i - base);
                const bool mv = rest & 1ull;
                const unsigned long long x = mv ? ~rest : rest;
                const int len = x ? __builtin_ctzll(x) : 64;
                const int e = (i + len < phi) ? i + len : phi;
                if (mv) span(forced, cut, std::true_type{}, i, e);
                else span(forced, cut, std::false_type{}, i, e);
                i = e;
            }
        };
        const int e0 = hi < M ? hi : M;               // forced rows end
        const int e1 = hi < cut_from ? hi : cut_from; // plain rows end
        phase(std::true_type{}, std::false_type{}, lo, e0);
        phase(std::false_type{}, std::false_type{}, lo > M ? lo : M, e1);
        phase(std::false_type{}, std::true_type{}, lo > cut_from ? lo : cut_from, hi);
    }

    // ---- outputs of the fill -----------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int j = sid[k];
        if (j == A.endstate && a.end_cost) a.end_cost[lr] = st.d[k];
        if (a.last_row && j >= 0) a.last_row[(size_t)lr * a.last_row_stride + j] = st.d[k];
    }
    if (lane == 0 && !a.check_status) a.status[lr] = 0;
}

// ------------------------------------------------------------------------------------------------
// General fill: any m >= 2, any fan-in <= 15, any S that fits the LDS ring.  A direct data-parallel
// statement of caller.py:217-244: the last m+1 rows of D live in an LDS ring, states are strided over
// the lanes, dwell sums are recomputed per candidate.  Slow path for unusual configurations.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void dtw_fill_generic(PassArgs a, int K)
{
    constexpr int PB = 4;
    constexpr int R = 32 / PB;
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63;
    const int slot = blockIdx.x;
    if (slot >= a.n_launch) return;
    ReadGeom gm = geom(a, slot);
    const int lr = rfl(gm.lr), T = rfl(gm.T);
    const long long off = gm.off;
    const DevAutomaton A = a.aut[a.aut_id[gm.r]];
    const int S = A.n_states;
    const int m = a.m;
    if (a.check_status && a.status[lr] != 0) return;
    if (T <= m || S <= m) {
        if (lane == 0) {
            a.status[lr] = 1;
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
    const long long boundary = (long long)A.flank_length - 10;
    const long long after_repeat = (long long)A.seq_idx_last - boundary;
    const long long first_threshold = 6 * boundary, second_threshold = (long long)T - 6 * boundary;
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
                }
            }
            bpw[k] |= ptr << ((i % R) * PB);
            // this row's inputs (rows i-1 and i-back) live in other ring rows
            row[j] = best;
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
}

// ------------------------------------------------------------------------------------------------
// Traceback: one thread per read.
//   bp words: word (wi, j) at bp[(wi*K + j/64)*64 + j%64], PB bits per row, R = 32/PB rows per word.
// Runs are appended in reverse time order: run_state[q], run_start[q]; adjacent equal states merge
// (self-loop states), matching the run-length encoding of the trace (caller.py:58-60).
// ------------------------------------------------------------------------------------------------
// ENC = 1: bit-per-predecessor fields, row r of a word at bits PB*(R-1-r) (dtw_fill_fast);
// ENC = 0: numeric pointer fields, row r at bits PB*r (dtw_fill_generic).
template <int PB, int ENC>
__global__ __launch_bounds__(64) void traceback_kernel(PassArgs a, int K)
{
    constexpr int R = 32 / PB;
    constexpr uint32_t PM = (1u << PB) - 1u;
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= a.n_launch) return;
    const ReadGeom gm = geom(a, slot);
    const int lr = gm.lr, T = gm.T;
    const long long off = gm.off;
    if (a.status[lr] != 0) {
        a.n_runs[lr] = 0;
        return;
    }
    const DevAutomaton &A = a.aut[a.aut_id[gm.r]];
    const int32_t *pred_ptr = A.pred_ptr, *pred_idx = A.pred_idx;
    const int m = a.m;
    const uint32_t *bp = a.bp + (size_t)(off / R + lr) * (K * 64);
    const uint32_t *maskw = a.maskbits ? (a.maskbits + (off / 32 + lr)) : nullptr;
    uint16_t *run_state = a.run_state + off;
    int32_t *run_start = a.run_start + off;
    const uint64_t *pred4 = A.pred4;
    const size_t stride = (size_t)K * 64;
    int j = A.endstate;
    int i = T - 1;
    int nr = 0;
    const uint16_t *pos = A.pos;
    // Run records are written 16 at a time: a 2-byte and a 4-byte store per run, from 64 lanes that each own a
    // different read, would miss in L2 on nearly every store (and turn into partial-sector HBM writes).  The open run
    // stays in registers (its start moves while the walk stays in the same state); finished runs queue up in LDS.
    __shared__ uint16_t q_state[16][64];
    __shared__ int32_t q_start[16][64];
    const int lane = threadIdx.x;
    int open_state = -1, open_start = 0;
    auto push = [&](int state, int start) {
        q_state[nr & 15][lane] = (uint16_t)state;
        q_start[nr & 15][lane] = start;
        nr++;
        if ((nr & 15) == 0) {
#pragma unroll
            for (int e = 0; e < 16; e++) run_state[nr - 16 + e] = q_state[e][lane];
#pragma unroll
            for (int e = 0; e < 16; e++) run_start[nr - 16 + e] = q_start[e][lane];
        }
    };
    while (true) {
        const int q = pos ? pos[j] : j; // where state j's pointer words live (slot q/64, lane q%64)
        const uint32_t *col = bp + (size_t)(q >> 6) * 64 + (q & 63);
        int wi = i / R;
        // two words per step (the current one and the one below) halve the dependent loads over runs of "stay"
        uint32_t wm = col[(size_t)wi * stride];
        uint32_t wn = wi > 0 ? col[(size_t)(wi - 1) * stride] : 0u;
        // keep rows <= i of this word
        if (ENC) {
            const int sh = (R - 1 - i % R) * PB; // bits below belong to later rows
            wm = (wm >> sh) << sh;
        } else {
            const int sh = (i % R + 1) * PB;
            if (sh < 32) wm &= (1u << sh) - 1u;
        }
        while (wm == 0 && wi > 0) {
            wi--;
            wm = wn;
            if (wm == 0 && wi > 0) {
                wi--;
                wm = col[(size_t)wi * stride];
                wn = wi > 0 ? col[(size_t)(wi - 1) * stride] : 0u;
            }
        }
        int start = 0, ptr = 0;
        if (wm != 0) {
            if (ENC) {
                const int fld = __builtin_ctz(wm) / PB;            // lowest non-zero field = latest row
                const uint32_t bits = (wm >> (fld * PB)) & PM;
                start = wi * R + (R - 1 - fld);
                ptr = 32 - __builtin_clz(bits);                     // highest set bit + 1
            } else {
                const int rr = (31 - __builtin_clz(wm)) / PB;
                start = wi * R + rr;
                ptr = (int)((wm >> (rr * PB)) & PM);
            }
        }
        if (j == open_state) {
            open_start = start;
        } else {
            if (open_state >= 0) push(open_state, open_start);
            open_state = j;
            open_start = start;
        }
        if (wm == 0) break;
        int back = m;
        if (maskw) back = ((maskw[start >> 5] >> (start & 31)) & 1u) ? m - 1 : m;
        if (ENC) j = (int)((pred4[j] >> (16 * (ptr - 1))) & 0xffffull); // fan-in <= 4 in the register-resident fill
        else j = pred_idx[pred_ptr[j] + ptr - 1];
        i = start - back;
        if (i < 0) break; // cannot happen: pointers are only set on rows >= m
    }
    if (open_state >= 0) push(open_state, open_start);
    for (int e = nr & ~15; e < nr; e++) {
        run_state[e] = q_state[e & 15][lane];
        run_start[e] = q_start[e & 15][lane];
    }
    a.n_runs[lr] = nr;
}

// per-sample state ids from the (reverse-ordered) run list; one wavefront per read
__global__ __launch_bounds__(256) void expand_trace_kernel(PassArgs a)
{
    const int lane = threadIdx.x & 63;
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= a.n_launch) return;
    const ReadGeom gm = geom(a, slot);
    if (a.status[gm.lr] != 0) return;
    const int n = a.n_runs[gm.lr];
    const uint16_t *rs = a.run_state + gm.off;
    const int32_t *rst = a.run_start + gm.off;
    uint16_t *trace = a.trace + gm.off;
    for (int q0 = 0; q0 < n; q0++) {
        const int start = rst[q0];
        const int end = (q0 == 0) ? gm.T : rst[q0 - 1]; // reverse order: the previous entry is the next run in time
        const uint16_t s = rs[q0];
        for (int q = start + lane; q < end; q += 64) trace[q] = s;
    }
}

template <int M, int K, int F, int FL>
hipError_t launch_fill(const PassArgs &a, hipStream_t s)
{
    const int blocks = (a.n_launch + 3) / 4;
    size_t shmem = 4 * (2 * (K * 64 + 1) + 136) * sizeof(double);
    // Occupancy cap (tuning knob): asking for more LDS per block leaves wave slots free for the latency-bound
    // kernels of other chunks that run beside the fill on other streams.
    static const int cap_blocks = [] {
        const char *e = getenv("WSX_FILL_BLOCKS_PER_CU");
        return e ? atoi(e) : 0;
    }();
    if (cap_blocks > 0) shmem = std::max(shmem, (size_t)(160 * 1024 / cap_blocks) & ~(size_t)255);
    if (shmem > 64 * 1024) shmem = 64 * 1024;
    hipLaunchKernelGGL((dtw_fill_fast<M, K, F, FL>), dim3(blocks), dim3(256), shmem, s, a);
    return hipGetLastError();
}

template <int M, int K>
hipError_t launch_fill_f(const PassArgs &a, int F, int FL, hipStream_t s)
{
    if constexpr (K >= 2 && M == 4) { // split variants (FL < F): several slots, default min_values_per_state
        if (F == 2 && FL == 1) return launch_fill<M, K, 2, 1>(a, s);
        if (F == 3 && FL == 1) return launch_fill<M, K, 3, 1>(a, s);
        if (F == 3 && FL == 2) return launch_fill<M, K, 3, 2>(a, s);
        if (F == 4 && FL == 1) return launch_fill<M, K, 4, 1>(a, s);
        if (F == 4 && FL == 2) return launch_fill<M, K, 4, 2>(a, s);
    }
    if (FL != F) return hipErrorInvalidValue;
    switch (F) {
    case 2: return launch_fill<M, K, 2, 2>(a, s);
    case 3: return launch_fill<M, K, 3, 3>(a, s);
    case 4: return launch_fill<M, K, 4, 4>(a, s);
    }
    return hipErrorInvalidValue;
}

template <int M>
hipError_t launch_fill_k(const PassArgs &a, int K, int F, int FL, hipStream_t s)
{
    switch (K) {
    case 1: return launch_fill_f<M, 1>(a, F, FL, s);
    case 2: return launch_fill_f<M, 2>(a, F, FL, s);
    case 3: return launch_fill_f<M, 3>(a, F, FL, s);
    case 4: return launch_fill_f<M, 4>(a, F, FL, s);
    case 5: return launch_fill_f<M, 5>(a, F, FL, s);
    }
    return hipErrorInvalidValue;
}

} // namespace

bool wsx_fast_pass_supported(int m, int K, int F)
{
    return m >= 3 && m <= 5 && K >= 1 && K <= WSX_MAX_K && F >= 1 && F <= WSX_MAX_F;
}

static int fast_f(int F) { return F <= 2 ? 2 : F; }

bool wsx_split_supported(int m, int K) { return m == 4 && K >= 2; }

const char *wsx_pass_kernel_name(int m, int K, int F, int FL, bool generic)
{
    static thread_local char buf[64];
    if (generic) snprintf(buf, sizeof(buf), "dtw_fill_generic");
    else snprintf(buf, sizeof(buf), "dtw_fill_fast<%d, %d, %d, %d>", m, K, fast_f(F), FL);
    return buf;
}

hipError_t wsx_launch_fill(const PassArgs &a, int m, int K, int F, int FL, bool generic, hipStream_t s)
{
#ifdef WSX_ONLY_DEFAULT // experiment builds: just the headline variant
    if (!generic && m == 4 && K == 1 && fast_f(F) == 2) return launch_fill<4, 1, 2, 2>(a, s);
    return hipErrorInvalidValue;
#else
    if (a.n_launch <= 0) return hipSuccess;
    if (generic) {
        const size_t shmem = (size_t)(m + 1) * K * 64 * sizeof(double);
        hipLaunchKernelGGL(dtw_fill_generic, dim3(a.n_launch), dim3(64), shmem, s, a, K);
        return hipGetLastError();
    }
    const int f = fast_f(F);
    switch (m) {
    case 3: return launch_fill_k<3>(a, K, f, FL, s);
    case 4: return launch_fill_k<4>(a, K, f, FL, s);
    case 5: return launch_fill_k<5>(a, K, f, FL, s);
    }
    return hipErrorInvalidValue;
#endif
}

hipError_t wsx_launch_traceback(const PassArgs &a, int K, int F, bool generic, hipStream_t s)
{
    if (a.n_launch <= 0) return hipSuccess;
    const int blocks = (a.n_launch + 63) / 64;
    if (generic) hipLaunchKernelGGL((traceback_kernel<4, 0>), dim3(blocks), dim3(64), 0, s, a, K);
    else if (fast_f(F) <= 2) hipLaunchKernelGGL((traceback_kernel<2, 1>), dim3(blocks), dim3(64), 0, s, a, K);
    else hipLaunchKernelGGL((traceback_kernel<4, 1>), dim3(blocks), dim3(64), 0, s, a, K);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !a.trace) return e;
    hipLaunchKernelGGL(expand_trace_kernel, dim3((a.n_launch + 3) / 4), dim3(256), 0, s, a);
    return hipGetLastError();
}
