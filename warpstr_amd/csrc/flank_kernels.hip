// flank_kernels.hip -- flank localisation (pipeline step 1 of upstream) on the GPU.
//
// What it computes (paths relative to the upstream repository):
//   find_sequence        src/extractor/tr_extractor.py:196-250  local alignment of a flank in the basecalled read
//                        (Bio.pairwise2.align.localms with alignment_config's scores, src/config.py:135-141) and the
//                        position / score / identity arithmetic on the aligned strings
//   transform_moves + extract_from_moves   tr_extractor.py:147-193   flank positions -> raw-signal positions
// PARITY UNPINNED against Biopython (absent here, no basecalled fixture upstream): the kernels are checked bit for bit
// against oracle/flank_oracle.c, which states the tie-breaking rules both share (end cell: best score, largest text
// index, then largest pattern index; traceback: diagonal, then text-base-against-gap, then pattern-base-against-gap,
// on the DP restricted to the last WR text rows).
//
// Kernels
//   flank_score_kernel<CPL>   one wavefront per (read, flank) pair.  Lane L owns pattern columns L*CPL+1 .. L*CPL+CPL
//       (CPL = 1, 2, 4: flanks up to 256 bases); the wave sweeps the text along anti-diagonals: at step t lane L is on
//       text row t - L, so the value it needs from its left neighbour was produced one step earlier and arrives with one
//       DPP move (wave_shr:1; lane 0 receives the zero of column 0).  Each lane fetches the eight text bases of its next
//       eight rows with one load.  No matrix is stored: only the best cell survives.  Integer max-plus work, VALU-bound
//       (7 VALU per cell + 3 per step); HBM traffic is the text once.
//   flank_trace_kernel<CPL>   one wavefront per pair: the same sweep over the last WR rows before the best cell with a
//       2-bit direction per cell parked in LDS (one byte per lane and row), then one lane walks back and evaluates
//       find_sequence's string arithmetic.  Tiny next to the score pass (WR <= 5p/3 + 2 rows).
//   moves_kernel              one 256-thread block per read: counts move-table entries per thread chunk, block prefix
//       sum, then each thread looks for the first block of context pos_start and the last of pos_end in its chunk.
//       Byte streaming, HBM-bound.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/warpstr_hip.h"

void wsx_internal_set_error(const char *msg);
extern "C" int wsx_internal_on_exception(void);

namespace {

#define FCHK(expr)                                                                                                \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess) {                                                                                   \
            char b_[512];                                                                                         \
            snprintf(b_, sizeof(b_), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);  \
            wsx_internal_set_error(b_);                                                                           \
            return WSX_ERR_HIP;                                                                                   \
        }                                                                                                         \
    } while (0)

struct FlankArgs {
    const uint8_t *text;
    const int64_t *text_off; // [n+1]
    const uint8_t *pat;
    const int64_t *pat_off;  // [n+1]
    int32_t n;
    int32_t match, mismatch, gap;
    // stage 1 -> stage 2
    int32_t *best;           // [n][4]: score, i, j, number of cells that reach the score
    wsx_flank_hit *hits;     // [n]
    uint8_t *ops;            // [n][ops_stride], ops_stride >= 2 * max pattern length + 8
    int32_t ops_stride;
};

__device__ __forceinline__ int rfl(int x) { return __builtin_amdgcn_readfirstlane(x); }

// One anti-diagonal step of the sweep for this lane's CPL columns.  left_in = the left neighbour's last column on the
// row this lane is on now (produced one step earlier), tc = the text base of that row.
template <int CPL, bool TRACE>
__device__ __forceinline__ void sweep_cell(const int (&pc)[CPL], int (&up)[CPL], int &diag_in, int left_in, int tc, bool valid,
                                           int match, int mismatch, int gap, int (&h_out)[CPL], uint32_t &dirs)
{
    int diag = diag_in, left = left_in;
    dirs = 0;
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const int hd = diag + (tc == pc[c] ? match : mismatch);
        const int hu = up[c] + gap;
        const int hl = left + gap;
        int h = max(max(hd, hu), max(hl, 0));
        if (!valid) h = 0;
        if (TRACE) { // diagonal first, then up (text base against a gap), then left; 3 = stop (score 0)
            const uint32_t d = h == 0 ? 3u : (h == hd ? 0u : (h == hu ? 1u : 2u));
            dirs |= d << (2 * c);
            // bit 8 + c: more than one predecessor reproduces the score (wsx_flank_hit::tie_steps counts them on the path)
            if (h != 0 && (int)(h == hd) + (int)(h == hu) + (int)(h == hl) > 1) dirs |= 0x100u << c;
        }
        diag = up[c];
        up[c] = h;
        left = h;
        h_out[c] = h;
    }
    diag_in = left_in; // the neighbour's value on this row is the diagonal input of the next row
}

// lane L receives lane L-1's value, lane 0 receives 0 (= column 0 of the matrix): one DPP move, no LDS round trip
__device__ __forceinline__ int from_left_lane(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x138 /* wave_shr:1 */, 0xf, 0xf, true); }

template <int CPL>
__global__ __launch_bounds__(256) void flank_score_kernel(FlankArgs a)
{
    const int lane = threadIdx.x & 63;
    const int w = rfl(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (w >= a.n) return;
    const long long to = a.text_off[w], po = a.pat_off[w];
    const int n = (int)(a.text_off[w + 1] - to), p = (int)(a.pat_off[w + 1] - po);
    const uint8_t *text = a.text + to, *pat = a.pat + po;
    int pc[CPL], up[CPL], h[CPL];
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const int j = lane * CPL + c; // 0-based column
        pc[c] = j < p ? (int)pat[j] : 0x100; // never equals a text base
        up[c] = 0;
    }
    int vmatch = a.match, vmismatch = a.mismatch; // kept in vector registers: v_cndmask cannot take two scalar operands
    asm volatile("" : "+v"(vmatch), "+v"(vmismatch));
    int best = 0, bi = 0, bj = 0; // this lane's latest cell that reached the wave's best score
    int nbest = 0;                // cells of this lane that reached `best`
    int wbest = 1;                // best score of the whole wave so far (uniform), at least 1: zeros never count
    int diag_in = 0, hlast = 0;
    // Eight steps of the sweep: steps t0 .. t0+7, this lane on text rows t0 - lane .. t0 - lane + 7, whose bases arrive
    // as one 8-byte load.  EDGE: groups in which some lane is above or below the text.
    auto group = [&](int t0, auto edge) {
        constexpr bool EDGE = decltype(edge)::value;
        uint32_t cw[2] = {0, 0};
        const int first = t0 - lane - 1; // 0-based text index of this lane's row at step t0
        if (!EDGE) {
            __builtin_memcpy(cw, text + first, 8);
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int idx = first + k;
                const uint32_t b = (idx >= 0 && idx < n) ? text[idx] : 0u;
                cw[k >> 2] |= b << (8 * (k & 3));
            }
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int tc = (int)((cw[k >> 2] >> (8 * (k & 3))) & 0xffu);
            const int in = from_left_lane(hlast); // the neighbour's last column on this lane's row
            const int i = t0 + k - lane;          // this lane's text row (1-based)
            const bool valid = !EDGE || (i >= 1 && i <= n);
            uint32_t dirs;
            sweep_cell<CPL, false>(pc, up, diag_in, in, tc, valid, vmatch, vmismatch, a.gap, h, dirs);
            // best cell: only a cell that reaches the best score of the whole wave so far can be (or tie with) the final
            // best cell, and such cells are rare -- the bookkeeping sits behind one compare per column and a branch
            bool any = false;
#pragma unroll
            for (int c = 0; c < CPL; c++) any |= h[c] >= wbest;
            if (__ballot(any) != 0ull) {
                int hm = 0;
#pragma unroll
                for (int c = 0; c < CPL; c++) {
                    const int j = lane * CPL + c + 1;
                    if (valid && j <= p && h[c] > 0 && h[c] >= wbest && h[c] >= best) { // later cells win ties within a lane
                        nbest = h[c] > best ? 1 : nbest + 1;
                        best = h[c];
                        bi = i;
                        bj = j;
                        hm = h[c];
                    }
                }
#pragma unroll
                for (int sft = 32; sft >= 1; sft >>= 1) hm = max(hm, __shfl_xor(hm, sft));
                wbest = max(wbest, rfl(hm));
            }
            hlast = h[CPL - 1];
        }
    };
    const int steps = n + 63;
    for (int t0 = 1; t0 <= steps; t0 += 8) {
        if (t0 >= 64 && t0 + 7 <= n) group(t0, std::false_type{});
        else group(t0, std::true_type{});
    }
    // best over the lanes: score, then text index, then pattern index
    unsigned long long key = ((unsigned long long)(unsigned)best << 48) | ((unsigned long long)(unsigned)bi << 16) | (unsigned)bj;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const unsigned long long o = __shfl_xor(key, s);
        key = o > key ? o : key;
    }
    // how many cells reach the best score (the lanes whose own best is the wave's)
    int cells = best == (int)(key >> 48) ? nbest : 0;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) cells += __shfl_xor(cells, s);
    if (lane == 0) {
        a.best[4 * w + 3] = cells;
        a.best[4 * w + 0] = (int)(key >> 48);
        a.best[4 * w + 1] = (int)((key >> 16) & 0xffffffffull);
        a.best[4 * w + 2] = (int)(key & 0xffffull);
    }
}

template <int CPL>
__global__ __launch_bounds__(64) void flank_trace_kernel(FlankArgs a, int max_rows)
{
    extern __shared__ uint8_t dirs_lds[]; // [rows + 1][64]: byte of lane L on window row r = directions of its CPL cells;
                                          // then [rows + 1][64] again: bit c = cell c of the lane has tied predecessors
    uint8_t *tie_lds = dirs_lds + (size_t)(max_rows + 1) * 64;
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x;
    if (w >= a.n) return;
    const long long to = a.text_off[w], po = a.pat_off[w];
    const int n = (int)(a.text_off[w + 1] - to), p = (int)(a.pat_off[w + 1] - po);
    const uint8_t *text = a.text + to, *pat = a.pat + po;
    wsx_flank_hit *hit = a.hits + w;
    const int best = a.best[4 * w], bi = a.best[4 * w + 1], bj = a.best[4 * w + 2];
    if (best <= 0) {
        if (lane == 0) {
            wsx_flank_hit z{};
            z.status = 1; // no positive-scoring alignment: pairwise2 returns [], upstream raises IndexError
            z.start = z.end = -1;
            *hit = z;
        }
        return;
    }
    const int wr = p + (a.match * p) / (-a.gap) + 2;
    const int i0 = bi - wr > 0 ? bi - wr : 0;
    const int rows = bi - i0; // <= max_rows
    int pc[CPL], up[CPL], h[CPL];
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const int j = lane * CPL + c;
        pc[c] = j < p ? (int)pat[j] : 0x100;
        up[c] = 0;
    }
    int diag_in = 0, hlast = 0;
    for (int t = 1; t <= rows + 63; t++) {
        const int r = t - lane; // window row 1..rows
        const bool valid = r >= 1 && r <= rows;
        const int tc = valid ? (int)text[i0 + r - 1] : 0;
        const int in = from_left_lane(hlast);
        uint32_t dirs;
        sweep_cell<CPL, true>(pc, up, diag_in, in, tc, valid, a.match, a.mismatch, a.gap, h, dirs);
        if (valid) {
            dirs_lds[r * 64 + lane] = (uint8_t)dirs;
            tie_lds[r * 64 + lane] = (uint8_t)(dirs >> 8);
        }
        hlast = h[CPL - 1];
    }
    __syncthreads();
    if (lane != 0) return;
    // ---- walk back from (bi, bj) ----
    int i = bi, j = bj, nops = 0, g1 = 0, g2 = 0, ties = 0;
    uint8_t *ops = a.ops + (size_t)w * a.ops_stride; // always present (the caller's buffer or a temporary), >= 2p + 8 bytes
    while (i > i0 && j > 0) {
        const int col = j - 1;
        const uint32_t d = (dirs_lds[(i - i0) * 64 + col / CPL] >> (2 * (col % CPL))) & 3u;
        if (d == 3u) break;
        ties += (tie_lds[(i - i0) * 64 + col / CPL] >> (col % CPL)) & 1;
        uint8_t op;
        if (d == 0u) {
            op = 'M';
            i--;
            j--;
        } else if (d == 1u) {
            op = 'U';
            i--;
            g2++;
        } else {
            op = 'L';
            j--;
            g1++;
        }
        ops[nops++] = op; // reverse order for now; nops <= p + match*p/(-gap) < ops_stride
    }
    for (int x = 0, y = nops - 1; x < y; x++, y--) {
        const uint8_t tmp = ops[x];
        ops[x] = ops[y];
        ops[y] = tmp;
    }
    for (int x = nops; x < a.ops_stride; x++) ops[x] = 0;
    // ---- find_sequence's arithmetic on the aligned strings (tr_extractor.py:226-250) ----
    const int lead_text = j > i ? j - i : 0, lead_pat = i > j ? i - j : 0;
    const int begin = i > j ? i : j;
    const int real_start = lead_pat;
    const int end = real_start + p + g2 - g1;
    const int suf_t = n - bi, suf_p = p - bj;
    const int alen = begin + nops + (suf_t > suf_p ? suf_t : suf_p);
    const int lo = real_start < alen ? real_start : alen;
    const int hi = end < alen ? (end > lo ? end : lo) : alen;
    int matches = 0;
    int ti = 0, pj = 0; // text / pattern bases of the local region consumed before the element being visited
    for (int q = 0; q < lo - begin && q < nops; q++) {
        if (ops[q] != 'L') ti++;
        if (ops[q] != 'U') pj++;
    }
    for (int t = lo; t < hi; t++) {
        int ca = -1, cb = -1; // -1 = gap character
        if (t < begin) {
            if (t >= lead_text) ca = text[t - lead_text];
            if (t >= lead_pat) cb = pat[t - lead_pat];
        } else if (t < begin + nops) {
            const uint8_t op = ops[t - begin];
            if (op != 'L') ca = text[i + ti++];
            if (op != 'U') cb = pat[j + pj++];
        } else {
            const int u = t - begin - nops;
            if (u < suf_t) ca = text[bi + u];
            if (u < suf_p) cb = pat[bj + u];
        }
        if (ca >= 0 && ca == cb) matches++;
    }
    wsx_flank_hit out{};
    out.status = 0;
    out.score = best + (p - ((hi - lo) - g2)) * a.gap;
    out.start = real_start;
    out.end = end;
    out.matches = matches;
    out.span = hi - lo;
    out.row0 = i;
    out.col0 = j;
    out.row1 = bi;
    out.col1 = bj;
    out.gaps_text = g1;
    out.gaps_pattern = g2;
    out.raw_score = best;
    out.n_ops = nops;
    out.n_best_cells = a.best[4 * w + 3];
    out.tie_steps = ties;
    *hit = out;
}

struct MovesArgs {
    const uint8_t *moves;
    const int64_t *off; // [n+1]
    const int32_t *pos_start, *pos_end;
    const int64_t *strand_start;
    const int32_t *block_stride;
    int64_t *raw_start, *raw_end;
    int32_t n;
};

// transform_moves + extract_from_moves: ctx(k) = #{1 <= q <= k : moves[q] != 0}; first k with ctx == pos_start, last k
// with ctx == pos_end
__global__ __launch_bounds__(256) void moves_kernel(MovesArgs a)
{
    __shared__ int part[256];
    __shared__ unsigned long long first_k, last_k; // k + 1; 0 = none
    const int r = blockIdx.x, tid = threadIdx.x;
    if (r >= a.n) return;
    const long long o = a.off[r];
    const long long m = a.off[r + 1] - o;
    const uint8_t *mv = a.moves + o;
    const long long per = (m + 255) / 256;
    const long long k0 = tid * per, k1 = k0 + per < m ? k0 + per : m;
    int cnt = 0;
    for (long long k = k0; k < k1; k++) cnt += (k > 0 && mv[k]) ? 1 : 0;
    part[tid] = cnt;
    if (tid == 0) {
        first_k = ~0ull;
        last_k = 0ull;
    }
    __syncthreads();
    // exclusive prefix over the 256 chunk counts (Hillis-Steele in LDS)
    for (int s = 1; s < 256; s <<= 1) {
        const int v = tid >= s ? part[tid - s] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int ctx = part[tid] - cnt; // context index before this chunk's first element
    const int ps = a.pos_start[r], pe = a.pos_end[r];
    long long f = -1, l = -1;
    for (long long k = k0; k < k1; k++) {
        if (k > 0 && mv[k]) ctx++;
        if (ctx == ps && f < 0) f = k;
        if (ctx == pe) l = k;
    }
    if (f >= 0) atomicMin(&first_k, (unsigned long long)f + 1ull);
    if (l >= 0) atomicMax(&last_k, (unsigned long long)l + 1ull);
    __syncthreads();
    if (tid == 0) {
        a.raw_start[r] = first_k != ~0ull ? a.strand_start[r] + (long long)(first_k - 1ull) * a.block_stride[r] : -1;
        a.raw_end[r] = last_k != 0ull ? a.strand_start[r] + (long long)(last_k - 1ull) * a.block_stride[r] : -1;
    }
}

struct DevTmp {
    std::vector<void *> ptrs;
    ~DevTmp()
    {
        for (void *p : ptrs) (void)hipFree(p);
    }
    hipError_t alloc(void **p, size_t bytes)
    {
        hipError_t e = hipMalloc(p, bytes ? bytes : 1);
        if (e == hipSuccess) ptrs.push_back(*p);
        return e;
    }
};

} // namespace

extern "C" {

int wsx_locate_flanks(int device, void *stream, int mem, const uint8_t *text, const int64_t *text_offsets,
                      const uint8_t *pattern, const int64_t *pattern_offsets, int64_t n, const wsx_align_scores *scores,
                      wsx_flank_hit *hits, uint8_t *ops, int32_t ops_stride)
try {
    if (!text_offsets || !pattern_offsets || !scores || !hits || n < 0 || (n > 0 && (!text || !pattern))) {
        wsx_internal_set_error("wsx_locate_flanks: null argument");
        return WSX_ERR_INVALID;
    }
    if (mem != WSX_MEM_HOST && mem != WSX_MEM_DEVICE) {
        wsx_internal_set_error("mem must be WSX_MEM_HOST or WSX_MEM_DEVICE");
        return WSX_ERR_INVALID;
    }
    if (scores->gap_open != scores->gap_extend || scores->gap_open >= 0 || scores->match <= 0 || scores->mismatch > 0) {
        wsx_internal_set_error("wsx_locate_flanks: needs gap_open == gap_extend < 0, match > 0, mismatch <= 0 (upstream: 2/-3/-3/-3)");
        return WSX_ERR_UNSUPPORTED;
    }
    if (n == 0) return WSX_SUCCESS;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        wsx_internal_set_error("no HIP device available");
        return WSX_ERR_NO_DEVICE;
    }
    FCHK(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    int max_p = 0;
    for (int64_t r = 0; r < n; r++) {
        const int64_t p = pattern_offsets[r + 1] - pattern_offsets[r], t = text_offsets[r + 1] - text_offsets[r];
        if (p <= 0 || p > 256 || t < 0 || t > (1 << 30)) {
            wsx_internal_set_error("wsx_locate_flanks: patterns must have 1..256 bases, texts at most 2^30");
            return WSX_ERR_INVALID;
        }
        max_p = std::max<int>(max_p, (int)p);
    }
    if (scores->match * max_p > 0x7fff) {
        wsx_internal_set_error("wsx_locate_flanks: scores do not fit 15 bits");
        return WSX_ERR_UNSUPPORTED;
    }
    const int64_t text_bytes = text_offsets[n] - text_offsets[0], pat_bytes = pattern_offsets[n] - pattern_offsets[0];
    DevTmp tmp;
    FlankArgs a{};
    int64_t *d_toff, *d_poff;
    FCHK(tmp.alloc((void **)&d_toff, (n + 1) * 8));
    FCHK(tmp.alloc((void **)&d_poff, (n + 1) * 8));
    std::vector<int64_t> toff(n + 1), poff(n + 1); // rebased to the start of the staged buffers
    for (int64_t r = 0; r <= n; r++) {
        toff[r] = text_offsets[r] - text_offsets[0];
        poff[r] = pattern_offsets[r] - pattern_offsets[0];
    }
    FCHK(hipMemcpyAsync(d_toff, toff.data(), (n + 1) * 8, hipMemcpyHostToDevice, st));
    FCHK(hipMemcpyAsync(d_poff, poff.data(), (n + 1) * 8, hipMemcpyHostToDevice, st));
    const bool host = mem == WSX_MEM_HOST;
    uint8_t *d_text = nullptr, *d_pat = nullptr, *d_ops = nullptr;
    wsx_flank_hit *d_hits = nullptr;
    if (host) {
        FCHK(tmp.alloc((void **)&d_text, text_bytes + 64));
        FCHK(tmp.alloc((void **)&d_pat, pat_bytes + 64));
        FCHK(tmp.alloc((void **)&d_hits, n * sizeof(wsx_flank_hit)));
        FCHK(hipMemcpyAsync(d_text, text + text_offsets[0], text_bytes, hipMemcpyHostToDevice, st));
        FCHK(hipMemcpyAsync(d_pat, pattern + pattern_offsets[0], pat_bytes, hipMemcpyHostToDevice, st));
    } else {
        d_text = const_cast<uint8_t *>(text) + text_offsets[0];
        d_pat = const_cast<uint8_t *>(pattern) + pattern_offsets[0];
        d_hits = hits;
        d_ops = ops;
    }
    const int32_t need_stride = 2 * max_p + 8;
    if (ops && ops_stride < need_stride) {
        wsx_internal_set_error("wsx_locate_flanks: ops_stride must be at least 2 * (longest pattern) + 8");
        return WSX_ERR_INVALID;
    }
    const int32_t stride = ops ? ops_stride : need_stride;
    if (host || !ops) FCHK(tmp.alloc((void **)&d_ops, (size_t)n * stride)); // identity needs the operations either way
    FCHK(tmp.alloc((void **)&a.best, n * 4 * sizeof(int32_t)));
    a.text = d_text;
    a.text_off = d_toff;
    a.pat = d_pat;
    a.pat_off = d_poff;
    a.n = (int32_t)n;
    a.match = scores->match;
    a.mismatch = scores->mismatch;
    a.gap = scores->gap_open;
    a.hits = d_hits;
    a.ops = d_ops;
    a.ops_stride = stride;
    const int cpl = max_p <= 64 ? 1 : (max_p <= 128 ? 2 : 4);
    const int max_rows = max_p + (scores->match * max_p) / (-scores->gap_open) + 2;
    const size_t lds = (size_t)(max_rows + 1) * 64 * 2; // directions + tie bits
    if (lds > 64 * 1024) { // (a launch that asks for more fails with a generic error: say what it is instead)
        wsx_internal_set_error("wsx_locate_flanks: the traceback window of the longest pattern does not fit 64 KB of LDS with these scores "
                               "(rows = pattern + match * pattern / |gap| + 2; upstream's 2 / -3 scores need 171 rows per 100 bases)");
        return WSX_ERR_UNSUPPORTED;
    }
    const dim3 g1((unsigned)((n + 3) / 4)), g2((unsigned)n);
    switch (cpl) {
    case 1:
        hipLaunchKernelGGL(flank_score_kernel<1>, g1, dim3(256), 0, st, a);
        hipLaunchKernelGGL(flank_trace_kernel<1>, g2, dim3(64), lds, st, a, max_rows);
        break;
    case 2:
        hipLaunchKernelGGL(flank_score_kernel<2>, g1, dim3(256), 0, st, a);
        hipLaunchKernelGGL(flank_trace_kernel<2>, g2, dim3(64), lds, st, a, max_rows);
        break;
    default:
        hipLaunchKernelGGL(flank_score_kernel<4>, g1, dim3(256), 0, st, a);
        hipLaunchKernelGGL(flank_trace_kernel<4>, g2, dim3(64), lds, st, a, max_rows);
        break;
    }
    FCHK(hipGetLastError());
    if (host) {
        FCHK(hipMemcpyAsync(hits, d_hits, n * sizeof(wsx_flank_hit), hipMemcpyDeviceToHost, st));
        if (ops) FCHK(hipMemcpyAsync(ops, d_ops, (size_t)n * stride, hipMemcpyDeviceToHost, st));
    }
    FCHK(hipStreamSynchronize(st)); // temporaries are freed on return
    return WSX_SUCCESS;
} catch (...) {
    return wsx_internal_on_exception();
}

int wsx_moves_to_raw(int device, void *stream, int mem, const uint8_t *moves, const int64_t *move_offsets,
                     const int32_t *pos_start, const int32_t *pos_end, const int64_t *strand_start,
                     const int32_t *block_stride, int64_t n, int64_t *raw_start, int64_t *raw_end)
try {
    if (!move_offsets || !pos_start || !pos_end || !strand_start || !block_stride || !raw_start || !raw_end || n < 0 ||
        (n > 0 && !moves)) {
        wsx_internal_set_error("wsx_moves_to_raw: null argument");
        return WSX_ERR_INVALID;
    }
    if (mem != WSX_MEM_HOST && mem != WSX_MEM_DEVICE) {
        wsx_internal_set_error("mem must be WSX_MEM_HOST or WSX_MEM_DEVICE");
        return WSX_ERR_INVALID;
    }
    if (n == 0) return WSX_SUCCESS;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        wsx_internal_set_error("no HIP device available");
        return WSX_ERR_NO_DEVICE;
    }
    FCHK(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    DevTmp tmp;
    MovesArgs a{};
    a.n = (int32_t)n;
    std::vector<int64_t> off(n + 1);
    for (int64_t r = 0; r <= n; r++) off[r] = move_offsets[r] - move_offsets[0];
    int64_t *d_off, *d_ss, *d_rs, *d_re;
    int32_t *d_ps, *d_pe, *d_bs;
    FCHK(tmp.alloc((void **)&d_off, (n + 1) * 8));
    FCHK(tmp.alloc((void **)&d_ss, n * 8));
    FCHK(tmp.alloc((void **)&d_ps, n * 4));
    FCHK(tmp.alloc((void **)&d_pe, n * 4));
    FCHK(tmp.alloc((void **)&d_bs, n * 4));
    FCHK(hipMemcpyAsync(d_off, off.data(), (n + 1) * 8, hipMemcpyHostToDevice, st));
    FCHK(hipMemcpyAsync(d_ss, strand_start, n * 8, hipMemcpyHostToDevice, st));
    FCHK(hipMemcpyAsync(d_ps, pos_start, n * 4, hipMemcpyHostToDevice, st));
    FCHK(hipMemcpyAsync(d_pe, pos_end, n * 4, hipMemcpyHostToDevice, st));
    FCHK(hipMemcpyAsync(d_bs, block_stride, n * 4, hipMemcpyHostToDevice, st));
    const bool host = mem == WSX_MEM_HOST;
    const int64_t bytes = move_offsets[n] - move_offsets[0];
    uint8_t *d_moves;
    if (host) {
        FCHK(tmp.alloc((void **)&d_moves, bytes + 64));
        FCHK(hipMemcpyAsync(d_moves, moves + move_offsets[0], bytes, hipMemcpyHostToDevice, st));
        FCHK(tmp.alloc((void **)&d_rs, n * 8));
        FCHK(tmp.alloc((void **)&d_re, n * 8));
    } else {
        d_moves = const_cast<uint8_t *>(moves) + move_offsets[0];
        d_rs = raw_start;
        d_re = raw_end;
    }
    a.moves = d_moves;
    a.off = d_off;
    a.pos_start = d_ps;
    a.pos_end = d_pe;
    a.strand_start = d_ss;
    a.block_stride = d_bs;
    a.raw_start = d_rs;
    a.raw_end = d_re;
    hipLaunchKernelGGL(moves_kernel, dim3((unsigned)n), dim3(256), 0, st, a);
    FCHK(hipGetLastError());
    if (host) {
        FCHK(hipMemcpyAsync(raw_start, d_rs, n * 8, hipMemcpyDeviceToHost, st));
        FCHK(hipMemcpyAsync(raw_end, d_re, n * 8, hipMemcpyDeviceToHost, st));
    }
    FCHK(hipStreamSynchronize(st));
    return WSX_SUCCESS;
} catch (...) {
    return wsx_internal_on_exception();
}

} // extern "C"
