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
//   dtw_fill_fast<M,K,F,FL,PK>  register-resident fill for min_values_per_state M in {3,4,5}: one 64-lane wavefront per
//       read; a state lives in one (slot, lane) of K slots per lane -- which one is decided by wsx_place.h (chains of
//       first-predecessor links, free of LDS bank conflicts; the natural "state j in lane j%64, slot j/64" when that is
//       as good); one row (= one signal sample) per step, all states of a row are independent.  Per state the "dwell" partial sums are a shift register
//       g_1..g_{M-1} that runs one row AHEAD of the DP:
//             after row i   g_s = D[i-s+1,j] + |s_{i-s+2}-v_j| + .. + |s_{i+1}-v_j|
//       so g_1 is the next row's stay candidate, and the value a successor needs at row i+2 is final at the end
//       of row i:  E_j(i+2) = g_{M-1} (unmasked row) or g_{M-2} (masked row).  E values are exchanged through LDS (one
//       8-byte slot per state, double buffered by row parity); a consumer's LDS reads for row i+1 are issued
//       during row i, a full row before they are needed.  Absent predecessors point at a slot holding +inf.
//       The signal arrives through the scalar data cache (s_load_dwordx16: eight samples per load, used as SGPR
//       operands).  Back-pointers never touch the vector ALU beyond the compare itself: "candidate f beat everything
//       before it" for the 64 states of a slot is one v_cmp_lt_f64 into an SGPR pair, and that pair -- one bit per
//       state -- is stored as it is with a scalar store (s_store_dwordx2/x4).  Per row the read gets NM = F + (K-1)*FL
//       64-bit masks in HBM; PK (one slot, two candidates, the states with two predecessors in lanes 0..7): the second
//       mask is one byte, 9 bytes per row in groups of 16 rows.
//   dtw_fill_generic            any m >= 2, fan-in <= 15: last m+1 rows of D in an LDS ring, direct restatement;
//       4-bit numeric pointers packed 8 rows per 32-bit word per state.
//   traceback_stream_kernel<F,PK>  mask layout, K = 1: one THREAD per read streams the read's mask rows downwards (the
//       rows visited do not depend on the path when there is one slot), tests its current state's bit per row.
//   traceback_mask_kernel<K,F,FL,PK> mask layout: one wavefront per read, 64 rows (all slots) per load, ballots.
//   traceback_generic_kernel    word layout of dtw_fill_generic, one thread per read.
//       All three emit the run-length state list in reverse time order.
//   expand_trace_kernel         optional: per-sample state ids from the run list (coalesced, wave per read).
//
// Roofline: min-plus recurrence, no MFMA.  Per row and 64 states (F = 2): 6 v_add_f64, 2 v_cmp_lt_f64, 2 v_min_f64 on the
// vector ALU; 2 ds_read_b64 + 1 ds_write_b64 on the LDS pipe (10 LDS cycles on gfx950, as many as the VALU needs: the two
// pipes are balanced); 1 scalar store.  HBM traffic per read and pass: 8T (signal) + 8*NM*T (masks, written once, read once
// by the traceback) + 6*runs; 8T + 9T + 6*runs with packed rows.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "wsx_device.h"

namespace {

constexpr double kInf = __builtin_huge_val();
#define WSX_AS4 __attribute__((address_space(4)))

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

// Back-pointer masks of a read: NM 64-bit masks per row, rows consecutive, first row 16-byte aligned.  Where a read's rows
// start is the host's decision (a.bp_off[lr], in 64-bit words: the reads of a chunk lie back to back whatever kernel variant
// each of them takes, wsx_api.hip: one region for all launch groups, sized by what the chunk's reads need).
__device__ __forceinline__ uint64_t *mask_rows(const PassArgs &a, long long off, int lr, int NM)
{
    return (uint64_t *)a.bp + a.bp_off[lr];
}

// Packed mask rows (PK; single-slot automata with two candidates whose states with two predecessors all sit in lanes 0..7,
// wsx_place.h): the second candidate's mask of a row is then one byte.  A read's rows are stored in groups of 16: sixteen
// 64-bit first-candidate masks followed by two 64-bit words that hold the sixteen bytes (row r of the group in byte r) --
// 144 bytes, so groups stay 16-byte aligned; 9 bytes per row instead of 16.
__device__ __forceinline__ uint64_t *mask_rows_pk(const PassArgs &a, long long off, int lr)
{
    return (uint64_t *)a.bp + a.bp_off[lr];
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
    bool cutf[K];
    double x[2][K]; // lane-major placement (LM): x[i&1][k] = the export slot k made at row i, for the state above it in the lane
    bool stk;       // LM = 4: this lane's slot WSX_DEV_STACK_SLOT starts a piece of its own (predecessor through LDS)
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

// Back-pointers are wave masks.  "cand < best-so-far" for all 64 states of a slot is ONE v_cmp_lt_f64 whose result is a
// 64-bit lane mask in an SGPR pair -- already the packed form (one bit per state), so no VALU op is spent on packing.
// The pair leaves the wave through the scalar memory pipe (s_store_dwordx2: SMEM issue, not VALU issue).
__device__ __forceinline__ uint64_t lt_mask(double cand, double best)
{
    return __builtin_amdgcn_fcmp(cand, best, 4 /* ordered less-than */);
}

template <int BYTE_OFF>
__device__ __forceinline__ void store_mask(uint64_t m, uint64_t *base)
{
    asm volatile("s_store_dwordx2 %0, %1, %2" ::"s"(m), "s"(base), "n"(BYTE_OFF) : "memory");
}

template <int BYTE_OFF> // two masks, 16-byte aligned destination
__device__ __forceinline__ void store_mask_pair(uint64_t m0, uint64_t m1, uint64_t *base)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = {(uint32_t)m0, (uint32_t)(m0 >> 32), (uint32_t)m1, (uint32_t)(m1 >> 32)};
    asm volatile("s_store_dwordx4 %0, %1, %2" ::"s"(v), "s"(base), "n"(BYTE_OFF) : "memory");
}

// the NM masks of group row R, at base + R*NM (base = the group's first row); pairs go out as one 16-byte store when
// the row size keeps them 16-byte aligned
template <int NM, int R, int Q = 0>
__device__ __forceinline__ void store_row_masks(const uint64_t (&mk)[NM], uint64_t *base)
{
    if constexpr (Q < NM) {
        if constexpr (NM % 2 == 0 && Q + 1 < NM) {
            store_mask_pair<(R * NM + Q) * 8>(mk[Q], mk[Q + 1], base);
            store_row_masks<NM, R, Q + 2>(mk, base);
        } else {
            store_mask<(R * NM + Q) * 8>(mk[Q], base);
            store_row_masks<NM, R, Q + 1>(mk, base);
        }
    }
}

// the masks of G consecutive rows (group rows RBASE .. RBASE+G-1), kept in SGPRs until now
template <int NM, int G, int RBASE, int Q = 0>
__device__ __forceinline__ void store_row_group(const uint64_t (&gm)[G][NM], uint64_t *base)
{
    if constexpr (Q < G) {
        store_row_masks<NM, RBASE + Q>(gm[Q], base);
        store_row_group<NM, G, RBASE, Q + 1>(gm, base);
    }
}

// masks per row: slot 0 holds F, every further slot FL; mask (k, f) is entry mask_index(k, f) of the row record
template <int F, int FL>
__host__ __device__ constexpr int mask_index(int k, int f) { return k == 0 ? f : F + (k - 1) * FL + f; }

// One DP row for all K slots.  PAR = row parity (selects LDS buffers and the e0/e1 roles at compile time),
// FORCED: rows 1..3 (D stays inf, only the pipeline advances), CUT: corner-cut rows.
//   top of row i : issue the LDS reads of E(i+1) (written at the end of row i-1) -- consumed in row i+1
//   body         : D[i,:] from the exports E(i) read one row earlier
//   end of row i : write E(i+2)
// Back-pointer encoding: per row, slot and candidate f one 64-bit mask (bit = lane): set <=> predecessor f beat everything
// before it in the reference's order (stay, pred 0, pred 1, ..); the arg-min is the highest f whose bit is set; none = stay.
// FL < F ("split"): only slot 0 considers F predecessors per state, the other slots FL (the host places every state
// with more than FL predecessors in slot 0; such states are few: loop entries, IUPAC alternatives).  FL == F: uniform.
// LM ("lane-major", wsx_place.h: wsx_place_lane_major): the state in (slot k >= 1, lane l) has ONE predecessor and it sits
// right below it, in (slot k-1, lane l) -- its candidate is a register of the same lane, two rows old (x[][]), not an LDS
// read; only slot 0 (chain heads, states with several predecessors, continuations of a chain from the lane before) reads
// LDS, and only the slots such states read from write it: slots 0 and K-1 (LM = 1), 0, 1 and K-1 (LM = 3) or all (LM = 2).  The slots are then
// walked downwards, so that slot k takes slot k-1's export of two rows ago before slot k-1 replaces it.
// LM = 4 ("stacked", wsx_place_lane_stacked): as LM = 2, and in the lanes of the automaton's stack mask the state in slot
// WSX_DEV_STACK_SLOT starts a piece of its own -- one predecessor, anywhere -- so that slot reads LDS as well and every lane
// picks its source (two v_cndmask per row).
template <int M, int K, int F, int FL, bool MROW, int PAR, bool FORCED, bool CUT, int LM = 0>
__device__ __forceinline__ void dp_row(FillState<M, K, F> &st, double *ex, int wl, double snext,
                                       uint64_t (&mk)[F + (K - 1) * FL], const uint64_t (&cutm)[K])
{
    static_assert(LM == 0 || (FL == 1 && K >= 2), "lane-major: one in-lane predecessor per state above slot 0");
    constexpr int EXW = K * 64 + 32;
    constexpr int wbuf = PAR * EXW;       // E(i+2) goes to the buffer of parity i
    constexpr int rbuf = (1 - PAR) * EXW; // E(i+1) lives in the buffer of parity i+1
#pragma unroll
    for (int k = 0; k < K; k++)
#pragma unroll
        for (int f = 0; f < F; f++) {
            if (k > 0 && (f >= FL || (LM != 0 && !(LM == 4 && k == WSX_DEV_STACK_SLOT)))) continue;
            const double e = ex[rbuf + st.paddr[k][f]];
            if (PAR) st.e0[k][f] = e;
            else st.e1[k][f] = e;
        }
#pragma unroll
    for (int kk = 0; kk < K; kk++) {
        const int k = LM != 0 ? K - 1 - kk : kk;
        double best = st.g[k][1];
        const int Fk = (k > 0) ? FL : F; // folds to a constant once the slot loop is unrolled
        if (FORCED) {
            best = kInf;
#pragma unroll
            for (int f = 0; f < F; f++)
                if (f < Fk) mk[mask_index<F, FL>(k, f)] = 0;
        } else {
            // candidates in the reference's order: stay, then the predecessors
#pragma unroll
            for (int f = 0; f < F; f++) {
                if (f < Fk) {
                    double src = (LM != 0 && k > 0) ? st.x[PAR][k - 1] : (PAR ? st.e1[k][f] : st.e0[k][f]);
                    if (LM == 4 && k == WSX_DEV_STACK_SLOT) src = st.stk ? (PAR ? st.e1[k][f] : st.e0[k][f]) : src;
                    const double cand = add_abs(src, st.acur[k]);
                    uint64_t lt = lt_mask(cand, best);
                    if (CUT) lt &= ~cutm[k];
                    mk[mask_index<F, FL>(k, f)] = lt;
                    best = min_f64(best, cand);
                }
            }
            if (CUT) best = st.cutf[k] ? kInf : best;
        }
        const double an = snext - st.v[k];
#pragma unroll
        for (int s = M - 1; s >= 2; s--) st.g[k][s] = add_abs(st.g[k][s - 1], an);
        st.g[k][1] = add_abs(best, an);
        st.d[k] = best;
        st.acur[k] = an;
        // row i+2 masked (back = M-1): successors need the (M-2)-deep sum, else the (M-1)-deep one
        // (wl: the lane's export slot -- its lane number, or for single-slot automata the slot wsx_place.h gave its state)
        const double enew = MROW ? st.g[k][M - 2] : st.g[k][M - 1];
        if (LM != 0 && k < K - 1) st.x[PAR][k] = enew;
        if (LM == 0 || LM == 2 || LM == 4 || k == 0 || k == K - 1 || (LM == 3 && k == 1)) ex[wbuf + (K == 1 ? wl : k * 64 + wl)] = enew;

    }
    __builtin_amdgcn_wave_barrier();
}

// (Register budgets: left to the compiler.  Forcing more waves per SIMD was measured slower every time -- two slots at 96
// registers no gain, four slots at 128 registers 6.0 vs 5.4 ms per launch, the lane-major four-slot kernel at 128 / 96 / 80
// registers 11.6 / 20.7 / 30.0 vs 10.35 ms per step: profiles/r02_occupancy_forcing.log, scripts/exp_round3_knobs.patch.
// Even a budget just below what the compiler takes buys nothing: the four-slot variants with three or four candidates use
// 172-190 registers (two waves); at 168 (three waves) they spill 20-84 values, none inside the 8-row blocks, parity stays green
// -- and DM2 at flank 110, whose 254-state strand runs <4,4,3,1>, takes 16.57 instead of 16.5-16.6 ms: profiles/r03_stacked_ab.log.)
template <int M, int K, int F, int FL, bool PK, int LM = 0>
#ifndef WSX_FILL_WPB
#define WSX_FILL_WPB 4 // wavefronts (= reads) per workgroup
#endif
__global__ __launch_bounds__(64 * WSX_FILL_WPB) void dtw_fill_fast(PassArgs a)
{
    static_assert(FL >= 1 && FL <= F, "slots 1.. consider FL <= F predecessors");
    static_assert(!PK || (K == 1 && F == 2), "packed mask rows: one slot, two candidates");
    static_assert(M >= 3, "the one-row-ahead export needs min_values_per_state >= 3");
    constexpr int NM = F + (K - 1) * FL; // back-pointer masks per row
    constexpr int EXW = K * 64 + 32; // export slots per buffer (+32 slots that hold +inf, one per bank pair)
    constexpr int WLDS = 2 * EXW; // doubles of LDS per wave: two export buffers
    extern __shared__ double lds[];

    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int slot = rfl(blockIdx.x * WSX_FILL_WPB + wib);
    if (slot >= a.n_launch) return;
    // the fill is the resource everything else waits for: its waves win issue arbitration against the latency-bound
    // stages of other chunks that share the SIMD (-1.5 % per step)
    __builtin_amdgcn_s_setprio(3);
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
    uint64_t *bp = PK ? mask_rows_pk(a, off, lr) : mask_rows(a, off, lr, NM); // row i's masks at bp[i*NM ..] (PK: see above)
    const int wl = K == 1 ? (int)A.wslot[lane] : lane; // export slot of this lane
    const uint32_t *maskw = a.maskbits ? (a.maskbits + (off / 32 + lr)) : nullptr;
    const int nmw = cdiv(T, 32);

    // ---- per-state constants -----------------------------------------------------------------
    FillState<M, K, F> st;
    const long long boundary = (long long)A.flank_length - 10;
    const long long after_repeat = (long long)A.seq_idx_last - boundary;
    long long cut_from_ll = 6 * boundary; // rows with i >= first_threshold and i > second_threshold
    if ((long long)T - 6 * boundary + 1 > cut_from_ll) cut_from_ll = (long long)T - 6 * boundary + 1;
    const int cut_from = rfl(cut_from_ll < M ? M : (cut_from_ll > T ? T : (int)cut_from_ll));
    // The cut only has to be FORCED for M rows: every predecessor of a cut state is a cut state (the cut takes a prefix of the
    // pattern: flank, repeat, the right flank's first bases; back edges stay inside the repeat), so once the values that were
    // in flight when the cut began have drained -- an export made at row i rests on D[i+2-M] at the oldest -- stay and
    // every candidate of a cut state are +inf by themselves, and so are its pointers ("inf < inf" is false).
    const int cut_end = cut_from + M < T ? cut_from + M : T;
    int sid[K]; // state handled at position k*64 + lane (-1: none)
    uint64_t cutm[K]; // lanes of slot k that the corner cut removes
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
        cutm[k] = __ballot(st.cutf[k]);
        // LDS slot of predecessor f's export; absent predecessors point at a +inf slot that the host chose so that the
        // read is free of bank conflicts (wsx_api.hip: fill_paddr)
#pragma unroll
        for (int f = 0; f < F; f++) st.paddr[k][f] = A.paddr[(k * WSX_MAX_F + f) * 64 + lane];
    }
    if (lane < 32) {
        ex[K * 64 + lane] = kInf;
        ex[EXW + K * 64 + lane] = kInf;
    }
    st.stk = LM == 4 && ((A.stack_mask >> lane) & 1ull);

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
        ex[0 * EXW + (K == 1 ? wl : k * 64 + lane)] = kInf; // E(2), E(1): never used (rows < M are forced to inf) but defined
        ex[1 * EXW + (K == 1 ? wl : k * 64 + lane)] = kInf;
#pragma unroll
        for (int f = 0; f < F; f++) {
            st.e0[k][f] = kInf;
            st.e1[k][f] = kInf;
        }
        st.x[0][k] = kInf;
        st.x[1][k] = kInf;
    }

    // Signal: the wave reads its samples through the scalar data cache.  The pointer is cast to the constant address
    // space, so the uniform loads below become s_load_dwordx16 -- eight samples per instruction, straight into SGPRs --
    // and the adds take them as scalar operands: no LDS traffic and no VALU op for the broadcast.  (The signal is
    // read-only for the whole launch.)  A vector load of the lines one block ahead only warms L2 for those scalar loads.
    typedef double d8 __attribute__((ext_vector_type(8)));
    typedef d8 d8u __attribute__((aligned(8)));
    const WSX_AS4 double *cs = (const WSX_AS4 double *)sig;
    const int last = T - 1;
    auto sample = [&](int q) -> double { return cs[q < T ? q : last]; };
    auto clampi = [&](int x) { return x < T ? x : last; };
    double warm = sig[clampi(64 + lane)];
    uint64_t acc = 0; // packed rows: the bytes of the current half group (rows 8h .. 8h+7) collected so far

    // The sample-mask words come through the scalar cache too (read-only during the fill): bit q of the read's mask =
    // sample q is masked; row r exports for row r+2, so its flag is bit r+2.
    const WSX_AS4 uint32_t *cm = (const WSX_AS4 uint32_t *)maskw;
    auto mask_word = [&](int w) -> uint32_t { return (cm && w < nmw) ? cm[w] : 0u; };
    // rows [i, e) share the flag of row i (e <= phi): scanned a mask word at a time, all scalar
    auto run_end = [&](int i, int phi, bool &mv) -> int {
        if (!cm) {
            mv = false;
            return phi;
        }
        int q = i + 2; // the sample whose mask bit is row i's flag
        uint32_t word = mask_word(q >> 5);
        mv = (word >> (q & 31)) & 1u;
        const int qend = phi + 2;
        while (q < qend) {
            const uint32_t x = (mv ? ~word : word) >> (q & 31); // bits that differ from the run's, from q on
            if (x != 0u) {
                q += __builtin_ctz(x);
                break;
            }
            q = (q | 31) + 1;
            word = mask_word(q >> 5);
        }
        return (q < qend ? q : qend) - 2;
    };
    // one row, everything wave-uniform except the per-lane state; snext = s_{i+1}
    auto row = [&](auto par, auto forced, auto cut, auto msk, auto rc, uint64_t *gp, double snext) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par)::value;
        constexpr int R = decltype(rc)::value; // row R of the group whose first row's masks are at gp
        constexpr bool FORCED = decltype(forced)::value;
        constexpr bool CUT = decltype(cut)::value;
        constexpr bool MROW = decltype(msk)::value;
        uint64_t mk[NM];
        dp_row<M, K, F, FL, MROW, PAR, FORCED, CUT, LM>(st, ex, wl, snext, mk, cutm);
        if (!FORCED) store_row_masks<NM, R>(mk, gp); // rows < M hold no pointers and are never read
    };
    // packed rows, one row: the first mask goes to its place in the group of 16, the second one's byte joins `acc`,
    // which leaves for HBM when the eighth row of its half group has been added
    auto row_pk = [&](auto par, auto forced, auto cut, auto msk, int r, double snext) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par)::value;
        constexpr bool FORCED = decltype(forced)::value;
        constexpr bool CUT = decltype(cut)::value;
        constexpr bool MROW = decltype(msk)::value;
        uint64_t mk[NM];
        dp_row<M, K, F, FL, MROW, PAR, FORCED, CUT, LM>(st, ex, wl, snext, mk, cutm);
        uint64_t *g16 = bp + (size_t)((unsigned)r >> 4) * 18;
        if (!FORCED) {
            store_mask<0>(mk[0], g16 + (r & 15));
            acc |= mk[NM - 1] << ((r & 7) * 8); // bits above 7 are never set: those lanes have one predecessor
        }
        if ((r & 7) == 7) {
            store_mask<0>(acc, g16 + 16 + ((r >> 3) & 1));
            acc = 0;
        }
    };
    // the same, the masks handed back instead of stored (the caller stores a whole group at once)
    auto row_keep = [&](auto par, auto forced, auto cut, auto msk, double snext, uint64_t (&mk)[NM]) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par)::value;
        constexpr bool FORCED = decltype(forced)::value;
        constexpr bool CUT = decltype(cut)::value;
        constexpr bool MROW = decltype(msk)::value;
        dp_row<M, K, F, FL, MROW, PAR, FORCED, CUT, LM>(st, ex, wl, snext, mk, cutm);
    };
    // rows [plo, phi) with constant compile-time flags: aligned groups of eight rows take their samples from one
    // 64-byte scalar load; the rows before and after such groups load theirs one by one
    auto span = [&](auto forced, auto cut, auto msk, int plo, int phi) __attribute__((always_inline)) {
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        using R0 = std::integral_constant<int, 0>;
        auto one = [&](int r) __attribute__((always_inline)) {
            if constexpr (PK) {
                if (r & 1) row_pk(P1{}, forced, cut, msk, r, sample(r + 1));
                else row_pk(P0{}, forced, cut, msk, r, sample(r + 1));
            } else {
                uint64_t *gp = bp + (size_t)(unsigned)r * NM;
                if (r & 1) row(P1{}, forced, cut, msk, R0{}, gp, sample(r + 1));
                else row(P0{}, forced, cut, msk, R0{}, gp, sample(r + 1));
            }
        };
        int i = plo;
        for (; i < phi && (i & 7); i++) one(i);
        for (; i + 8 <= phi && i + 8 < T; i += 8) {
            // once per 64 rows: the lines two blocks ahead, for the scalar loads (the earlier load has landed long ago: no
            // stall).  Without it the step is 4 % slower; eight lanes fetching a line every group instead, 2 % slower.
            if ((i & 63) == 0) {
                asm volatile("" ::"v"(warm));
                warm = sig[clampi(i + 128 + lane)];
            }
            const d8 v = *(const WSX_AS4 d8u *)(cs + i + 1);
            uint64_t *gp = bp + (size_t)(unsigned)i * NM;
            if constexpr (PK && !decltype(forced)::value) {
                // eight aligned rows (`acc` is empty here: it was flushed after row i-1): the first masks leave as four
                // 16-byte scalar stores, the eight bytes as one 8-byte store -- 72 bytes instead of 128
                uint64_t *g16 = bp + (size_t)((unsigned)i >> 4) * 18;
                uint64_t m0[4][2]; // rows 2q and 2q+1 side by side: one 16-byte store
                uint32_t b_lo = 0, b_hi = 0;
#define WSX_ROW(R)                                                                                                  \
{                                                                                                               \
    uint64_t mk[NM];                                                                                            \
    row_keep(std::integral_constant<int, (R)&1>{}, forced, cut, msk, v[R], mk);                                 \
    m0[(R) / 2][(R) % 2] = mk[0];                                                                               \
    if constexpr ((R) < 4) b_lo |= (uint32_t)mk[1] << (8 * (R));                                                \
    else b_hi |= (uint32_t)mk[1] << (8 * ((R)-4));                                                             \
}
                WSX_ROW(0)
                WSX_ROW(1)
                WSX_ROW(2)
                WSX_ROW(3)
                WSX_ROW(4)
                WSX_ROW(5)
                WSX_ROW(6)
                WSX_ROW(7)
#undef WSX_ROW
                uint64_t *gp8 = g16 + (i & 8);
                store_row_group<2, 4, 0>(m0, gp8);
                store_mask<0>(((uint64_t)b_hi << 32) | b_lo, g16 + 16 + ((i >> 3) & 1));
            } else if constexpr (PK) {
                for (int q = 0; q < 8; q++) one(i + q); // (forced rows: the first four rows of a read)
            } else if constexpr (NM <= 4 && !decltype(forced)::value) {
                // The scalar stores of G rows leave together (G = as many rows as fit ~32 SGPRs of masks; single-slot
                // automata: with more masks per row the grouping bought nothing).  A scalar
                // store counts on the same counter as the LDS reads and completes out of order with them, so a store
                // in flight turns every counted LDS wait behind it into a wait for the store as well.
                constexpr int G = NM <= 2 ? 8 : 4;
                uint64_t gm[G][NM];
#define WSX_ROW(R)                                                                                                  \
row_keep(std::integral_constant<int, (R)&1>{}, forced, cut, msk, v[R], gm[(R) % G]);                            \
if constexpr (((R) % G) == G - 1) store_row_group<NM, G, (R) - (G - 1)>(gm, gp);
                WSX_ROW(0)
                WSX_ROW(1)
                WSX_ROW(2)
                WSX_ROW(3)
                WSX_ROW(4)
                WSX_ROW(5)
                WSX_ROW(6)
                WSX_ROW(7)
#undef WSX_ROW
            } else {
                row(P0{}, forced, cut, msk, std::integral_constant<int, 0>{}, gp, v[0]);
                row(P1{}, forced, cut, msk, std::integral_constant<int, 1>{}, gp, v[1]);
                row(P0{}, forced, cut, msk, std::integral_constant<int, 2>{}, gp, v[2]);
                row(P1{}, forced, cut, msk, std::integral_constant<int, 3>{}, gp, v[3]);
                row(P0{}, forced, cut, msk, std::integral_constant<int, 4>{}, gp, v[4]);
                row(P1{}, forced, cut, msk, std::integral_constant<int, 5>{}, gp, v[5]);
                row(P0{}, forced, cut, msk, std::integral_constant<int, 6>{}, gp, v[6]);
                row(P1{}, forced, cut, msk, std::integral_constant<int, 7>{}, gp, v[7]);
            }
        }
        for (; i < phi; i++) one(i);
    };
    // A phase (forced rows, plain rows, the rows that force the corner cut, plain rows again) is cut into maximal runs of equal
    // mask flag, so that the row code is branch-free; a run is NOT cut at 64-row boundaries any more: every change from one
    // run's code to another's costs a shuffle of the whole state between the registers the two loops were allocated (26 moves
    // for four slots), which at one change per 64 rows was 0.5-1.1 vector instructions per row.
    auto phase = [&](auto forced, auto cut, int plo, int phi) __attribute__((always_inline)) {
        int i = plo;
        while (i < phi) {
            bool mv;
            const int e = run_end(i, phi, mv);
            if (mv) span(forced, cut, std::true_type{}, i, e);
            else span(forced, cut, std::false_type{}, i, e);
            i = e;
        }
    };
    {
        const int e0 = T < M ? T : M; // forced rows end
        phase(std::true_type{}, std::false_type{}, 1, e0);
        phase(std::false_type{}, std::false_type{}, M, cut_from);
        phase(std::false_type{}, std::true_type{}, cut_from, cut_end);
        phase(std::false_type{}, std::false_type{}, cut_end, T);
    }

    if constexpr (PK) { // the last half group of the read, if it is not complete
        if ((T & 7) != 0) store_mask<0>(acc, bp + (size_t)((unsigned)last >> 4) * 18 + 16 + ((last >> 3) & 1));
    }
    // ---- outputs of the fill -----------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int j = sid[k];
        if (j == A.endstate && a.end_cost) a.end_cost[lr] = st.d[k];
        if (a.last_row && j >= 0) a.last_row[(size_t)lr * a.last_row_stride + j] = st.d[k];
    }
    if (lane == 0 && !a.check_status) a.status[lr] = 0;
    asm volatile("s_dcache_wb" ::: "memory"); // scalar stores sit in the scalar data cache until written back
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
    uint32_t *bp = a.bp + 2 * a.bp_off[lr]; // (R rows per 32-bit word and state: (T / R + 1) * K * 64 words per read)
    const uint32_t *maskw = a.maskbits ? (a.maskbits + (off / 32 + lr)) : nullptr;
    const int ring = m + 1;
    const int SP = K * 64;
    // ring row q holds D[i, :] for i % ring == q
    for (int q = lane; q < ring * SP; q += 64) lds[q] = kInf;
    __builtin_amdgcn_wave_barrier();
    const double v0 = A.value[0];
    const double start_val = fabs(sig[0] - v0);
    if (lane == 0) lds[0] = start_val;
    for (int j = lane; j <= m && j < S; j += 64) // (min_values_per_state may exceed the 64 lanes)
        if (j >= 1) lds[j] = start_val + fabs(sig[j] - v0);
    const long long boundary = (long long)A.flank_length - 10;
    const long long after_repeat = (long long)A.seq_idx_last - boundary;
    const long long first_threshold = 6 * boundary, second_threshold = (long long)T - 6 * boundary;
    __builtin_amdgcn_wave_barrier();
    // the pointer words being filled (8 rows x 4 bits per state) live in LDS behind the ring: K is a run-time value here
    // (any number of slots the ring has room for), so a per-thread array would have no static size
    uint32_t *bpw = (uint32_t *)(lds + (size_t)ring * SP);
    for (int k = 0; k < K; k++) bpw[k * 64 + lane] = 0;
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
            bpw[k * 64 + lane] |= ptr << ((i % R) * PB);
            // this row's inputs (rows i-1 and i-back) live in other ring rows
            row[j] = best;
        }
        if ((i % R) == R - 1 || i == T - 1) {
            const int wi = i / R;
            for (int k = 0; k < K; k++) {
                bp[((size_t)wi * K + k) * 64 + lane] = bpw[k * 64 + lane];
                bpw[k * 64 + lane] = 0;
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
// Traceback over the mask layout of dtw_fill_fast: one wavefront per read.
//   row i of the read holds NM 64-bit masks at bp[i*NM + mask_index(k, f)]; bit = lane of the state in slot k.
// A block is 64 consecutive rows (64b .. 64b+63), one row per lane with ALL its NM masks: a coalesced load.  Whether
// the state at (slot, bit) was entered at a row is that row's lane testing bit `bit` of the slot's masks; a ballot turns
// the block into a 64-row history of the state and count-leading-zeros finds the latest entry at or below the current
// row.  Every transition inside the block's 64 rows is walked without touching memory, whichever slots the path visits
// (loop entries and other states with several predecessors sit in slot 0, so paths through long automata change slot at
// every repeat unit), and the two blocks below are already in flight while a block is walked.  The walk runs in position
// space (slot*64 + lane); predecessor positions (pred4) and state ids come through the scalar cache, so a transition
// costs the vector ALU two instructions per candidate.  Runs are appended in reverse time order: run_state[q],
// run_start[q]; adjacent equal states merge, matching the run-length encoding of the trace (caller.py:58-60).
// ------------------------------------------------------------------------------------------------
template <int K, int F, int FL, bool PK>
__global__ __launch_bounds__(256) void traceback_mask_kernel(PassArgs a)
{
    static_assert(!PK || (K == 1 && F == 2), "packed mask rows: one slot, two candidates");
    const int lane = threadIdx.x & 63;
    const int slot = rfl(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (slot >= a.n_launch) return;
    const ReadGeom gm = geom(a, slot);
    const int lr = rfl(gm.lr), T = rfl(gm.T);
    const long long off = gm.off;
    if (a.status[lr] != 0) {
        if (lane == 0) a.n_runs[lr] = 0;
        return;
    }
    const DevAutomaton &A = a.aut[a.aut_id[gm.r]];
    const int m = a.m;
    constexpr int NM = F + (K - 1) * FL;
    const uint64_t *bp = PK ? mask_rows_pk(a, off, lr) : mask_rows(a, off, lr, NM);
    const uint32_t *maskw = a.maskbits ? (a.maskbits + (off / 32 + lr)) : nullptr;
    uint16_t *run_state = a.run_state + off;
    int32_t *run_start = a.run_start + off;
    // predecessor positions by position: scalar loads (uniform index), no vector instruction
    const WSX_AS4 uint64_t *pred4 = (const WSX_AS4 uint64_t *)A.pred4;
    const uint16_t *state_at = A.state_at;
    int q = rfl(A.pos ? (int)A.pos[A.endstate] : A.endstate);
    int i = T - 1;
    int nr = 0;
    // finished runs (position, start row) queue up one per lane and leave 64 at a time (coalesced); positions turn into
    // state ids on the way out
    int q_pos = 0, q_start = 0;
    int open_pos = -1, open_start = 0;
    auto flush = [&](int first, int count) {
        if (lane < count) {
            run_state[first + lane] = state_at ? state_at[q_pos] : (uint16_t)q_pos;
            run_start[first + lane] = q_start;
        }
    };
    auto push = [&](int pos, int start) {
        if (lane == (nr & 63)) {
            q_pos = pos;
            q_start = start;
        }
        nr++;
        if ((nr & 63) == 0) flush(nr - 64, 64);
    };
    struct Block {
        uint64_t w[NM]; // the row's masks, index mask_index(k, f)
        uint32_t mw;    // the sample-mask word of this lane's row
    };
    auto load = [&](Block &B, int b) {
        const int row = b * 64 + lane;
        const bool in = b >= 0 && row < T;
        if constexpr (PK) { // groups of 16 rows: the row's first mask, and its byte of the second one
            const int rw = in ? row : 0;
            const uint64_t *g16 = bp + (size_t)(rw >> 4) * 18;
            const bool has = in && row >= m;
            B.w[0] = has ? g16[rw & 15] : 0ull;
            B.w[1] = has ? ((g16[16 + ((rw >> 3) & 1)] >> ((rw & 7) * 8)) & 0xffull) : 0ull;
        } else {
            const uint64_t *rp = bp + (size_t)(in ? row : 0) * NM;
#pragma unroll
            for (int e = 0; e < NM; e++) B.w[e] = (in && row >= m) ? rp[e] : 0ull; // rows < m hold no pointers
        }
        B.mw = (maskw && in) ? maskw[row >> 5] : 0u;
    };
    // Walks every transition inside block cb (rows 64*cb ..), starting at row i in position q.  Returns true when the
    // walk has reached row 0, false when it continues in the block below.
    auto walk = [&](const Block &cur, int cb) -> bool {
        const uint64_t cmasked = __ballot((cur.mw >> (lane & 31)) & 1u); // bit l: row 64*cb + l is masked (back = m-1)
        while (true) {
            const int ks = q >> 6, bit = q & 63;
            // won[f] bit l: predecessor f beat everything before it at row 64*cb + l -- a shift, an AND and a compare
            // per candidate on the vector ALU (the slot's masks are picked by a uniform register index); the rest is scalar
            // The slot is wave-uniform: a branch per slot picks that slot's masks by a STATIC register index (a shift, an AND
            // and a compare per candidate; a chain of selects over the K slots cost 2 (K-1) more per candidate).
            uint64_t won[F];
#pragma unroll
            for (int f = 0; f < F; f++) won[f] = 0ull;
            auto history = [&](uint64_t w) { return __ballot((w >> bit) & 1ull); };
            if (ks == 0) {
#pragma unroll
                for (int f = 0; f < F; f++) won[f] = history(cur.w[f]);
            }
#define WSX_TB_SLOT(KK)                                                                                             \
    if constexpr (K > KK) {                                                                                         \
        if (ks == KK) {                                                                                             \
            _Pragma("unroll") for (int f = 0; f < FL; f++) won[f] = history(cur.w[mask_index<F, FL>(KK, f)]);      \
        }                                                                                                           \
    }
            WSX_TB_SLOT(1)
            WSX_TB_SLOT(2)
            WSX_TB_SLOT(3)
            WSX_TB_SLOT(4)
#undef WSX_TB_SLOT
            static_assert(K <= 5, "one branch per slot");
            uint64_t entered = 0;
#pragma unroll
            for (int f = 0; f < F; f++) entered |= won[f];
            entered &= ~0ull >> (63 - (i & 63)); // rows <= i
            if (entered == 0 && cb * 64 > m) {   // the run continues in the block below
                i = cb * 64 - 1;
                return false;
            }
            int start = 0, ptr = 0, l = 0;
            if (entered != 0) {
                l = 63 - __builtin_clzll(entered);
                start = cb * 64 + l;
                ptr = 1; // the arg-min is the highest candidate whose bit is set
#pragma unroll
                for (int f = 1; f < F; f++) ptr = ((won[f] >> l) & 1ull) ? f + 1 : ptr;
            }
            if (q == open_pos) {
                open_start = start;
            } else {
                if (open_pos >= 0) push(open_pos, open_start);
                open_pos = q;
                open_start = start;
            }
            if (entered == 0) return true; // reached row 0 in this state
            const int back = m - (int)((cmasked >> l) & 1ull);
            // lane-major placement: the one predecessor of a position above slot 0 sits right below it -- no table look-up
            // (a dependent scalar load per transition otherwise)
            if (a.lane_major && ks > 0 && !(ks == WSX_DEV_STACK_SLOT && ((A.stack_mask >> bit) & 1ull))) q -= 64;
            else q = (int)((pred4[q] >> (16 * (ptr - 1))) & 0xffffull); // fan-in <= 4 in the register-resident fill
            i = start - back;
            if ((i >> 6) != cb) return false; // a step never skips a block: back <= m rows
        }
    };
    // three blocks in registers, roles rotating statically: while one is walked the two below it are in flight
    Block b0, b1, b2;
    int cb = i >> 6;
    load(b0, cb);
    load(b1, cb - 1);
    load(b2, cb - 2);
    while (true) {
        if (walk(b0, cb)) break;
        cb--;
        load(b0, cb - 2);
        if (walk(b1, cb)) break;
        cb--;
        load(b1, cb - 2);
        if (walk(b2, cb)) break;
        cb--;
        load(b2, cb - 2);
    }
    if (open_pos >= 0) push(open_pos, open_start);
    flush(nr & ~63, nr & 63);
    if (lane == 0) a.n_runs[lr] = nr;
}

// ------------------------------------------------------------------------------------------------
// Traceback over the mask layout for single-slot automata (K = 1, S <= 64): one THREAD per read.
// With one slot the rows a walk touches do not depend on the states it visits: the walk reads row T-1, T-2, .. of the
// read's masks, whatever the path.  Each thread therefore streams its read's rows downwards (RC rows = 96..128
// contiguous bytes per step, two steps in flight), tests the bit of its current state in each row and, where it is
// set, closes the run and moves to the predecessor.  Per row that is a handful of VALU instructions shared by 64
// reads; the stream runs at HBM rate (scripts/exp_stream.hip: 5.4 TB/s for this access pattern).  Predecessor
// positions come from a table in LDS (one 64-entry table per automaton of the handle); finished runs queue up in LDS
// and are written 16 at a time.  Same outputs as traceback_mask_kernel.
// (Tried: eight lanes fetching one read's 128-byte step together and trading pieces through LDS -- whole-line loads
// instead of 64 lines per instruction.  7 % faster alone, 4 % slower per step: the fill it runs beside is LDS-bound.)
// ------------------------------------------------------------------------------------------------
template <int F, bool PK>
__global__ __launch_bounds__(64) void traceback_stream_kernel(PassArgs a, int n_aut)
{
    static_assert(!PK || F == 2, "packed mask rows: two candidates");
    constexpr int RC = F == 2 ? 8 : 4; // rows per step
    typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
    extern __shared__ uint64_t tb_tab[]; // [automaton][position]: the positions of its state's predecessors; then the state ids
    constexpr int QD = 32; // queue depth per lane: a 16-entry chunk waiting to leave + what three steps can add (<= 3 * RC / 2)
    __shared__ uint16_t q_state[QD][64];
    __shared__ int32_t q_start[QD][64];
    const int lane = threadIdx.x;
    uint16_t *st_tab = (uint16_t *)(tb_tab + n_aut * 64); // [automaton][position] -> state id
    for (int e = lane; e < n_aut * 64; e += 64) {
        const DevAutomaton &B = a.aut[e >> 6];
        const bool one_slot = B.n_states <= 64;
        tb_tab[e] = one_slot ? B.pred4[e & 63] : 0ull;
        st_tab[e] = (one_slot && B.state_at) ? B.state_at[e & 63] : (uint16_t)(e & 63);
    }
    __syncthreads();
    const int slot = blockIdx.x * 64 + lane;
    if (slot >= a.n_launch) return;
    const ReadGeom gm = geom(a, slot);
    const int lr = gm.lr, T = gm.T;
    const long long off = gm.off;
    if (a.status[lr] != 0) {
        a.n_runs[lr] = 0;
        return;
    }
    const int aid = a.aut_id[gm.r];
    const uint64_t *tab = tb_tab + aid * 64;
    const uint16_t *stab = st_tab + aid * 64;
    const int m = a.m;
    const uint64_t *bp = PK ? mask_rows_pk(a, off, lr) : mask_rows(a, off, lr, F);
    const uint32_t *maskw = a.maskbits ? (a.maskbits + (off / 32 + lr)) : nullptr;
    uint16_t *run_state = a.run_state + off;
    int32_t *run_start = a.run_start + off;
    int nr = 0;
    int nf = 0; // runs already written out (a multiple of 16)
    auto push = [&](int position, int start) {
        q_state[nr & (QD - 1)][lane] = stab[position];
        q_start[nr & (QD - 1)][lane] = start;
        nr++;
    };
    // Full 16-entry chunks leave the queue here -- called between steps, not from push(): the lanes of a wave fill their
    // queues at different times, so a test inside the transition code made nearly every transition of the WAVE run the
    // (long) write-out for some lane.
    auto drain = [&]() {
        if (nr - nf >= 16) {
#pragma unroll
            for (int e = 0; e < 16; e++) run_state[nf + e] = q_state[(nf + e) & (QD - 1)][lane];
#pragma unroll
            for (int e = 0; e < 16; e++) run_start[nf + e] = q_start[(nf + e) & (QD - 1)][lane];
            nf += 16;
        }
    };
    int open_state = -1, open_start = 0; // (positions; they turn into state ids in push)
    auto close_run = [&](int state, int start) { // the walk leaves `state`, entered at row `start`
        if (state == open_state) {
            open_start = start;
        } else {
            if (open_state >= 0) push(open_state, open_start);
            open_state = state;
            open_start = start;
        }
    };
    struct Step {
        ull2 v[PK ? RC / 2 : RC * F / 2]; // RC rows x F masks (packed rows: RC first masks)
        unsigned long long pk;           // packed rows: the RC bytes of the second mask
        uint32_t mw;                     // the sample-mask word that covers these rows
    };
    uint32_t mw_word = 0u;
    int mw_at = -1;
    auto load = [&](Step &S, int c) { // rows RC*c .. RC*c + RC-1 (c < 0: nothing)
        const int cc = c < 0 ? 0 : c;
        if constexpr (PK) { // half of a 16-row group: 64 bytes of first masks + its 8 bytes of the word pair behind them
            const uint64_t *g16 = bp + (size_t)(cc >> 1) * 18;
            const ull2 *p = (const ull2 *)(g16 + (cc & 1) * 8);
#pragma unroll
            for (int e = 0; e < RC / 2; e++) S.v[e] = p[e];
            S.pk = g16[16 + (cc & 1)];
        } else {
            const ull2 *p = (const ull2 *)(bp + (size_t)cc * (RC * F));
#pragma unroll
            for (int e = 0; e < RC * F / 2; e++) S.v[e] = p[e];
            S.pk = 0ull;
        }
        // the sample-mask word changes every 32 / RC steps only (steps are loaded in descending order)
        const int wi = (cc * RC) >> 5;
        if (maskw && wi != mw_at) {
            mw_word = maskw[wi];
            mw_at = wi;
        }
        S.mw = mw_word;
    };
    int bit = a.aut[aid].pos ? (int)a.aut[aid].pos[a.aut[aid].endstate] : a.aut[aid].endstate; // the walk's position (= lane of the fill)
    int i = T - 1;                 // rows above i are not part of the walk
    int c = i / RC;
    // three steps in registers with statically rotating roles: one is processed while the loads of the other two are in
    // flight (copying a step from one set of registers to another would wait for its load)
    Step s0, s1, s2;
    load(s0, c);
    load(s1, c - 1);
    load(s2, c - 2);
    // A step is walked transition by transition, not row by row: for the walk's current position every lane gathers, for
    // each candidate f, the RC rows' bits into a small vector (three VALU ops per row and candidate), keeps the rows the
    // walk can still reach (<= i, >= m) and takes the highest one.  With ~12 rows per run nearly every ROW sees some
    // lane of the wave change state, so a per-row branch made every lane pay the transition code RC times per step; this
    // way it runs as often as the busiest lane changes state inside the step (two or three times).
    auto process = [&](const Step &cur, int cc) {
        const int base = cc * RC;
        while (true) {
            const int top = i - base; // rows base .. base+top are still part of the walk
            uint32_t reach = top >= RC - 1 ? (1u << RC) - 1u : (top < 0 ? 0u : (2u << top) - 1u);
            if (base < m) reach &= ~((1u << (m - base)) - 1u); // rows < m hold no pointers (and were never written)
            const uint32_t sh = (uint32_t)bit & 31u;
            const bool upper = bit >= 32;
            uint32_t hf[F], h = 0;
#pragma unroll
            for (int f = 0; f < F; f++) {
                hf[f] = 0;
                if (PK && f == 1) {
                    // the second candidate's bytes: bit `bit` of byte rr -> bit rr (positions >= 8 have one predecessor)
                    if (bit < 8) {
                        unsigned long long t = (cur.pk >> bit) & 0x0101010101010101ull;
                        t |= t >> 7;
                        t |= t >> 14;
                        t |= t >> 28;
                        hf[f] = (uint32_t)t & 0xffu;
                    }
                } else {
#pragma unroll
                    for (int rr = 0; rr < RC; rr++) {
                        const int idx = PK ? rr : rr * F + f;
                        const unsigned long long w = (idx & 1) ? cur.v[idx / 2].y : cur.v[idx / 2].x;
                        const uint32_t d = upper ? (uint32_t)(w >> 32) : (uint32_t)w;
                        // bit `sh` of d lands at bit rr: one bit-field extract and one shift-or (the compiler's own choice
                        // was shift, shift, and, or)
                        uint32_t t;
                        asm("v_bfe_u32 %0, %1, %2, 1" : "=v"(t) : "v"(d), "v"(sh));
                        asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(hf[f]) : "v"(t), "n"(rr), "v"(hf[f]));
                    }
                }
                h |= hf[f];
            }
            h &= reach;
            if (__ballot(h != 0) == 0) break;
            if (h != 0) {
                const int rr = 31 - __builtin_clz(h); // the latest reachable row where the state was entered
                const int r = base + rr;
                close_run(bit, r);
                int ptr = 1; // the arg-min is the highest candidate whose bit is set
#pragma unroll
                for (int f = 1; f < F; f++) ptr = ((hf[f] >> rr) & 1u) ? f + 1 : ptr;
                const int back = m - (int)((cur.mw >> (r & 31)) & 1u);
                bit = (int)((tab[bit] >> (16 * (ptr - 1))) & 0xffffull);
                i = r - back;
            }
        }
    };
    // (steps below row 0 are empty: their loads are clamped and `r <= i` is false for their rows)
    for (; c >= 0; c -= 3) {
        process(s0, c);
        load(s0, c - 3);
        process(s1, c - 1);
        load(s1, c - 4);
        process(s2, c - 2);
        load(s2, c - 5);
        drain();
    }
    close_run(bit, 0); // row 0 is reached in this state
    if (open_state >= 0) push(open_state, open_start);
    drain();
    for (int e = nf; e < nr; e++) {
        run_state[e] = q_state[e & (QD - 1)][lane];
        run_start[e] = q_start[e & (QD - 1)][lane];
    }
    a.n_runs[lr] = nr;
}

// ------------------------------------------------------------------------------------------------
// Traceback over the word layout of dtw_fill_generic: one thread per read.
//   bp words: word (wi, j) at bp[(wi*K + j/64)*64 + j%64], 4 bits per row (numeric pointer, row r at bits 4*r),
//   8 rows per word.  Same outputs as above.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void traceback_generic_kernel(PassArgs a, int K)
{
    constexpr int PB = 4;
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
    const uint32_t *bp = a.bp + 2 * a.bp_off[lr];
    const uint32_t *maskw = a.maskbits ? (a.maskbits + (off / 32 + lr)) : nullptr;
    uint16_t *run_state = a.run_state + off;
    int32_t *run_start = a.run_start + off;
    const size_t stride = (size_t)K * 64;
    int j = A.endstate;
    int i = T - 1;
    int nr = 0;
    int open_state = -1, open_start = 0;
    auto push = [&](int state, int start) {
        run_state[nr] = (uint16_t)state;
        run_start[nr] = start;
        nr++;
    };
    while (true) {
        const uint32_t *col = bp + (size_t)(j >> 6) * 64 + (j & 63);
        int wi = i / R;
        uint32_t wm = col[(size_t)wi * stride];
        const int sh = (i % R + 1) * PB; // keep rows <= i of this word
        if (sh < 32) wm &= (1u << sh) - 1u;
        while (wm == 0 && wi > 0) {
            wi--;
            wm = col[(size_t)wi * stride];
        }
        int start = 0, ptr = 0;
        if (wm != 0) {
            const int rr = (31 - __builtin_clz(wm)) / PB;
            start = wi * R + rr;
            ptr = (int)((wm >> (rr * PB)) & PM);
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
        j = pred_idx[pred_ptr[j] + ptr - 1];
        i = start - back;
        if (i < 0) break; // cannot happen: pointers are only set on rows >= m
    }
    if (open_state >= 0) push(open_state, open_start);
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

template <int M, int K, int F, int FL, bool PK = false, int LM = 0>
hipError_t launch_fill(const PassArgs &a, const WsxTuning &tun, hipStream_t s)
{
    const int blocks = (a.n_launch + WSX_FILL_WPB - 1) / WSX_FILL_WPB;
    static_assert(WSX_FILL_WPB * (2 * (K * 64 + 32)) * sizeof(double) <= 64 * 1024, "export buffers of a workgroup: at most 64 KB of LDS");
    size_t shmem = WSX_FILL_WPB * (2 * (K * 64 + 32)) * sizeof(double);
    // Occupancy cap (tuning knob): asking for more LDS per block leaves wave slots free for the latency-bound
    // kernels of other chunks that run beside the fill on other streams.
    const int cap_blocks = tun.fill_blocks_per_cu;
    if (cap_blocks > 0) shmem = std::max(shmem, (size_t)(160 * 1024 / cap_blocks) & ~(size_t)255);
    if (shmem > 64 * 1024) shmem = 64 * 1024;
    hipLaunchKernelGGL((dtw_fill_fast<M, K, F, FL, PK, LM>), dim3(blocks), dim3(64 * WSX_FILL_WPB), shmem, s, a);
    return hipGetLastError();
}

template <int M, int K>
hipError_t launch_fill_f(const PassArgs &a, int F, int FL, bool pk, int lm, const WsxTuning &tun, hipStream_t s)
{
    if (lm != 0) { // lane-major placement: slots above 0 take their one predecessor from the lane's own registers
        if constexpr (K >= 2 && M == 4) {
            if (FL != 1 || pk || lm > 4) return hipErrorInvalidValue;
            if constexpr (K >= 4) {
                switch (F * 10 + lm) {
                case 23: return launch_fill<M, K, 2, 1, false, 3>(a, tun, s);
                case 33: return launch_fill<M, K, 3, 1, false, 3>(a, tun, s);
                case 43: return launch_fill<M, K, 4, 1, false, 3>(a, tun, s);
                }
            }
#ifdef WSX_EXPERIMENT
            constexpr int kStackedMinK = 4; // (experiment builds: the four-slot stacked variants too, WSX_STACKED_MIN_K=4)
#else
            constexpr int kStackedMinK = 5;
#endif
            if constexpr (K >= kStackedMinK) { // stacked placement: five slots only (four: measured slower than slot-major, wsx_api.hip)
                switch (F * 10 + lm) {
                case 24: return launch_fill<M, K, 2, 1, false, 4>(a, tun, s);
                case 34: return launch_fill<M, K, 3, 1, false, 4>(a, tun, s);
                case 44: return launch_fill<M, K, 4, 1, false, 4>(a, tun, s);
                }
            }
            if (lm == 4) return hipErrorInvalidValue;
            switch (F * 10 + (lm == 3 ? 2 : lm)) {
            case 21: return launch_fill<M, K, 2, 1, false, 1>(a, tun, s);
            case 22: return launch_fill<M, K, 2, 1, false, 2>(a, tun, s);
            case 31: return launch_fill<M, K, 3, 1, false, 1>(a, tun, s);
            case 32: return launch_fill<M, K, 3, 1, false, 2>(a, tun, s);
            case 41: return launch_fill<M, K, 4, 1, false, 1>(a, tun, s);
            case 42: return launch_fill<M, K, 4, 1, false, 2>(a, tun, s);
            }
        }
        return hipErrorInvalidValue;
    }
    if constexpr (K >= 2 && M == 4) { // split variants (FL < F): several slots, default min_values_per_state
        if (F == 2 && FL == 1) return launch_fill<M, K, 2, 1>(a, tun, s);
        if (F == 3 && FL == 1) return launch_fill<M, K, 3, 1>(a, tun, s);
        if (F == 3 && FL == 2) return launch_fill<M, K, 3, 2>(a, tun, s);
        if (F == 4 && FL == 1) return launch_fill<M, K, 4, 1>(a, tun, s);
        if (F == 4 && FL == 2) return launch_fill<M, K, 4, 2>(a, tun, s);
    }
    if (FL != F) return hipErrorInvalidValue;
    if constexpr (K == 1) {
        if (pk && F == 2) return launch_fill<M, 1, 2, 2, true>(a, tun, s);
    }
    if (pk) return hipErrorInvalidValue;
    switch (F) {
    case 2: return launch_fill<M, K, 2, 2>(a, tun, s);
    case 3: return launch_fill<M, K, 3, 3>(a, tun, s);
    case 4: return launch_fill<M, K, 4, 4>(a, tun, s);
    }
    return hipErrorInvalidValue;
}

template <int M>
hipError_t launch_fill_k(const PassArgs &a, int K, int F, int FL, bool pk, int lm, const WsxTuning &tun, hipStream_t s)
{
    switch (K) {
    case 1: return launch_fill_f<M, 1>(a, F, FL, pk, lm, tun, s);
    case 2: return launch_fill_f<M, 2>(a, F, FL, pk, lm, tun, s);
    case 3: return launch_fill_f<M, 3>(a, F, FL, pk, lm, tun, s);
    case 4: return launch_fill_f<M, 4>(a, F, FL, pk, lm, tun, s);
    case 5: return launch_fill_f<M, 5>(a, F, FL, pk, lm, tun, s);
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

// Lane-major placement from three slots on: with two it buys 2-4 % where it fits (4.30 vs 4.46 ms at 65 states, 4.71 vs 4.82
// at 96: profiles/r03_state_staircase.log) and costs the mixed batch of configs[4] 4 % (more kernel variants per call:
// 16.9 vs 16.2 ms per step).  WSX_FILL_LM=3 asks for it from two slots on.
bool wsx_lane_major_supported(int m, int K)
{
    static const int lm_min_k = [] {
        const char *e = wsx_exp_env("WSX_FILL_LM");
        return (e && atoi(e) == 3) ? 2 : 3;
    }();
    return m == 4 && K >= lm_min_k && K <= WSX_MAX_K;
}

const char *wsx_pass_kernel_name(int m, int K, int F, int FL, bool pk, int lm, bool generic)
{
    static thread_local char buf[64];
    if (generic) snprintf(buf, sizeof(buf), "dtw_fill_generic");
    else snprintf(buf, sizeof(buf), "dtw_fill_fast<%d, %d, %d, %d, %s, %d>", m, K, fast_f(F), FL, pk ? "true" : "false", lm); // as rocprofv3 prints it
    return buf;
}

hipError_t wsx_launch_fill(const PassArgs &a, int m, int K, int F, int FL, bool pk, int lm, bool generic, const WsxTuning &tun, hipStream_t s)
{
#ifdef WSX_ONLY_WG // listing builds (hipcc -S): just the flank-110 variants of the several-slot fills
    return lm ? launch_fill<4, 4, 2, 1, false, 1>(a, tun, s) : launch_fill<4, 4, 2, 1>(a, tun, s);
#elif defined(WSX_ONLY_DEFAULT) // experiment builds: just the headline variant
    if (!generic && m == 4 && K == 1 && fast_f(F) == 2) return pk ? launch_fill<4, 1, 2, 2, true>(a, tun, s) : launch_fill<4, 1, 2, 2>(a, tun, s);
    return hipErrorInvalidValue;
#else
    if (a.n_launch <= 0) return hipSuccess;
    if (generic) {
        const size_t shmem = (size_t)(m + 1) * K * 64 * sizeof(double) + (size_t)K * 64 * sizeof(uint32_t);
        hipLaunchKernelGGL(dtw_fill_generic, dim3(a.n_launch), dim3(64), shmem, s, a, K);
        return hipGetLastError();
    }
    const int f = fast_f(F);
    switch (m) {
    case 3: return launch_fill_k<3>(a, K, f, FL, pk, lm, tun, s);
    case 4: return launch_fill_k<4>(a, K, f, FL, pk, lm, tun, s);
    case 5: return launch_fill_k<5>(a, K, f, FL, pk, lm, tun, s);
    }
    return hipErrorInvalidValue;
#endif
}

hipError_t wsx_launch_expand_trace(const PassArgs &a, hipStream_t s) // per-sample state ids from the run list (every traceback's epilogue)
{
    if (a.n_launch <= 0 || !a.trace) return hipSuccess;
    hipLaunchKernelGGL(expand_trace_kernel, dim3((a.n_launch + 3) / 4), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t wsx_launch_traceback(const PassArgs &a, int K, int F, int FL, bool pk, bool generic, int n_aut, const WsxTuning &tun, hipStream_t s)
{
    if (a.n_launch <= 0) return hipSuccess;
    const int wblocks = (a.n_launch + 3) / 4;
    const int stream_min = tun.stream_traceback_min; // smallest launch that takes the streaming traceback
    if (generic) {
        hipLaunchKernelGGL(traceback_generic_kernel, dim3((a.n_launch + 63) / 64), dim3(64), 0, s, a, K);
    } else if (K == 1 && n_aut <= 64 && a.n_launch >= stream_min) {
        // thread per read; one 512-byte table per automaton in LDS.  Fewest instructions per read, but a launch lasts as
        // long as one thread needs for its whole read (~0.3 ms): small launches take the wave-per-read kernel instead
        // (measured: 1k reads 0.77 vs 1.45 ms per call, 16k reads 2.7 vs 3.2 ms; from 12.5k reads per launch on,
        // inside a 100k-read batch, the streaming kernel wins)
        const int tblocks = (a.n_launch + 63) / 64;
        const size_t shmem = (size_t)n_aut * 64 * (sizeof(uint64_t) + sizeof(uint16_t));
        const int f = fast_f(F);
        if (pk && f != 2) return hipErrorInvalidValue;
        if (pk) hipLaunchKernelGGL((traceback_stream_kernel<2, true>), dim3(tblocks), dim3(64), shmem, s, a, n_aut);
        else if (f == 2) hipLaunchKernelGGL((traceback_stream_kernel<2, false>), dim3(tblocks), dim3(64), shmem, s, a, n_aut);
        else if (f == 3) hipLaunchKernelGGL((traceback_stream_kernel<3, false>), dim3(tblocks), dim3(64), shmem, s, a, n_aut);
        else if (f == 4) hipLaunchKernelGGL((traceback_stream_kernel<4, false>), dim3(tblocks), dim3(64), shmem, s, a, n_aut);
        else return hipErrorInvalidValue;
    } else {
        const int f = fast_f(F);
#define WSX_TB(KK, FF, LL)                                                                                          \
    if (K == KK && f == FF && FL == LL && !pk)                                                                      \
    hipLaunchKernelGGL((traceback_mask_kernel<KK, FF, LL, false>), dim3(wblocks), dim3(256), 0, s, a)
#define WSX_TB_SPLIT(KK)                                                                                            \
    WSX_TB(KK, 2, 2);                                                                                               \
    else WSX_TB(KK, 3, 3);                                                                                          \
    else WSX_TB(KK, 4, 4);                                                                                          \
    else WSX_TB(KK, 2, 1);                                                                                          \
    else WSX_TB(KK, 3, 1);                                                                                          \
    else WSX_TB(KK, 3, 2);                                                                                          \
    else WSX_TB(KK, 4, 1);                                                                                          \
    else WSX_TB(KK, 4, 2)
        if (K == 1 && f == 2 && FL == 2 && pk)
            hipLaunchKernelGGL((traceback_mask_kernel<1, 2, 2, true>), dim3(wblocks), dim3(256), 0, s, a);
        else WSX_TB(1, 2, 2);
        else WSX_TB(1, 3, 3);
        else WSX_TB(1, 4, 4);
        else WSX_TB_SPLIT(2);
        else WSX_TB_SPLIT(3);
        else WSX_TB_SPLIT(4);
        else WSX_TB_SPLIT(5);
        else return hipErrorInvalidValue;
#undef WSX_TB_SPLIT
#undef WSX_TB
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !a.trace) return e;
    hipLaunchKernelGGL(expand_trace_kernel, dim3((a.n_launch + 3) / 4), dim3(256), 0, s, a);
    return hipGetLastError();
}
