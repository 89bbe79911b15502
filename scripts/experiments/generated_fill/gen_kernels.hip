// gen_kernels.hip -- what the library itself brings to the GENERATED fills (warpstr_amd/fillgen.py writes and compiles the
// fill kernels per automaton; wsx_caller_set_generated_fill attaches them): the traceback over their back-pointer words.
#include "wsx_device.h"

namespace {

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
// Traceback over the words of a GENERATED fill (warpstr_amd/fillgen.py: a read in four lanes, 16 reads per wavefront): one
// thread per read.  Row i of a wave holds gen_nwp 64-bit words; word w is the compare mask of one add / compare / min group
// of the generated code, bit (g*4 + q) belongs to lane q of the wave's g-th read.  For the walk's position the automaton's
// table names the words of its candidates (in `incoming` order) and the predecessors' positions; the arg-min is the highest
// candidate whose bit is set, none = stay -- the rule of the other tracebacks.  Eight rows of the position's words are fetched
// at a time; the walk is taken transition by transition.  Same outputs as traceback_stream_kernel.
// ------------------------------------------------------------------------------------------------
// One wavefront per wavefront of the fill (16 reads): its rows come down in stages of TBT_ROWS rows, fetched by all 64 lanes
// (a stage is contiguous memory) into LDS, the next stage's words waiting in registers meanwhile; lanes 0..15 walk one read
// each through the stage -- the bits of a position's candidates over eight rows per step, taken transition by transition as
// in traceback_stream_kernel, but out of LDS, where a dependent step costs a hundred cycles instead of a trip to HBM.
constexpr int TBT_ROWS = 32;
__global__ __launch_bounds__(64) void traceback_t_kernel(PassArgs a, int nwp)
{
    extern __shared__ uint64_t tbt_lds[]; // [2][TBT_ROWS * nwp] row words, then the automaton's tables
    const int lane = threadIdx.x;
    const int slot0 = blockIdx.x * WSX_GEN_RPW;
    const int r0 = a.order[slot0]; // the wave's longest read; all reads of a launch group share the automaton
    const DevAutomaton &A = a.aut[a.aut_id[r0]];
    const int n = A.gen_n, P = 4 * n, m = a.m;
    const int Tmax = (int)(a.offsets[r0 + 1] - a.offsets[r0]);
    const uint64_t *bp = (const uint64_t *)a.bp + a.bp_off[r0 - a.first_read];
    const int stage_words = TBT_ROWS * nwp;
    uint16_t *t_word = (uint16_t *)(tbt_lds + 2 * stage_words), *t_pred = t_word + P * WSX_MAX_F, *t_state = t_pred + P * WSX_MAX_F;
    uint8_t *t_n = (uint8_t *)(t_state + P);
    for (int e = lane; e < P * WSX_MAX_F; e += 64) {
        t_word[e] = A.gen_tb_word[e];
        t_pred[e] = A.gen_tb_pred[e];
    }
    for (int e = lane; e < P; e += 64) {
        t_state[e] = A.gen_state_at[e];
        t_n[e] = A.gen_tb_n[e];
    }
    // ---- the walkers ----
    const int slot = slot0 + lane;
    bool active = lane < WSX_GEN_RPW && slot < a.n_launch;
    int lr = 0, T = 0;
    long long off = 0;
    if (active) {
        const ReadGeom gm = geom(a, slot);
        lr = gm.lr;
        T = gm.T;
        off = gm.off;
        if (a.status[lr] != 0) {
            a.n_runs[lr] = 0;
            active = false;
        }
    }
    const uint32_t *maskw = a.maskbits ? (a.maskbits + (off / 32 + lr)) : nullptr;
    uint16_t *run_state = a.run_state + off;
    int32_t *run_start = a.run_start + off;
    int nr = 0, open_pos = -1, open_start = 0;
    auto close_run = [&](int p, int start) { // the walk leaves position p, entered at row `start`
        if (p == open_pos) {
            open_start = start;
        } else {
            if (open_pos >= 0) {
                run_state[nr] = t_state[open_pos];
                run_start[nr] = open_start;
                nr++;
            }
            open_pos = p;
            open_start = start;
        }
    };
    int p = A.gen_end_pos;
    int i = T - 1;
    const int sh0 = lane * 4; // bit of lane q of this read: g*4 + q
    // ---- stages, top down ----
    const int n_stage = (Tmax + TBT_ROWS - 1) / TBT_ROWS;
    constexpr int PER = 16; // words per lane and stage held in registers (TBT_ROWS * nwp <= 64 * PER)
    uint64_t hold[PER];
    auto fetch = [&](int st) { // stage st -> registers (rows beyond the region's end are inside its slack)
#pragma unroll
        for (int e = 0; e < PER; e++) {
            const int w = lane + 64 * e;
            hold[e] = (st >= 0 && w < stage_words) ? bp[(size_t)st * stage_words + w] : 0ull;
        }
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int e = 0; e < PER; e++) {
            const int w = lane + 64 * e;
            if (w < stage_words) tbt_lds[buf * stage_words + w] = hold[e];
        }
    };
    fetch(n_stage - 1);
    park(0);
    __syncthreads();
    for (int st = n_stage - 1, buf = 0; st >= 0; st--, buf ^= 1) {
        fetch(st - 1); // in flight while the walkers are busy
        const int base = st * TBT_ROWS;
        const uint64_t *rows = tbt_lds + buf * stage_words;
        while (active && i >= base && i >= m) {
            const int blk = i & ~7;
            const int nf = t_n[p];
            const int sh = sh0 + p / n;
            uint32_t hf[WSX_MAX_F] = {0, 0, 0, 0}, h = 0;
            const int top = i - blk; // rows blk .. blk + top are part of the walk
            uint32_t reach = (2u << top) - 1u;
            if (blk < m) reach &= ~((1u << (m - blk)) - 1u); // rows < m hold no pointers (and were never written)
#pragma unroll
            for (int f = 0; f < WSX_MAX_F; f++) {
                if (f < nf) {
                    const uint64_t *wp = rows + (size_t)(blk - base) * nwp + t_word[p * WSX_MAX_F + f];
#pragma unroll
                    for (int rr = 0; rr < 8; rr++) hf[f] |= (uint32_t)((wp[rr * nwp] >> sh) & 1ull) << rr;
                    hf[f] &= reach;
                    h |= hf[f];
                }
            }
            if (h == 0) {
                i = blk - 1;
                continue;
            }
            const int rr = 31 - __builtin_clz(h); // the latest reachable row where the state was entered
            const int r = blk + rr;
            close_run(p, r);
            int ptr = 0; // the arg-min is the highest candidate whose bit is set
#pragma unroll
            for (int f = 0; f < WSX_MAX_F; f++) ptr = (f < nf && ((hf[f] >> rr) & 1u)) ? f : ptr;
            const int back = m - (maskw ? (int)((maskw[r >> 5] >> (r & 31)) & 1u) : 0);
            p = t_pred[p * WSX_MAX_F + ptr];
            i = r - back;
        }
        __syncthreads(); // every walker has left this buffer's predecessor
        park(buf ^ 1);
        __syncthreads();
    }
    if (active) {
        close_run(p, 0); // row 0 is reached in this state
        if (open_pos >= 0) {
            run_state[nr] = t_state[open_pos];
            run_start[nr] = open_start;
            nr++;
        }
        a.n_runs[lr] = nr;
    }
}

} // namespace

hipError_t wsx_launch_traceback_t(const PassArgs &a, int nwp, int n_per_lane, hipStream_t s)
{
    if (a.n_launch <= 0) return hipSuccess;
    if (TBT_ROWS * nwp > 64 * 16) return hipErrorInvalidValue; // (a stage must fit the lanes' holding registers: nwp <= 32)
    const int P = 4 * n_per_lane;
    const size_t shmem = (size_t)2 * TBT_ROWS * nwp * 8 + (size_t)P * WSX_MAX_F * 4 + (size_t)P * 2 + (size_t)P + 16;
    hipLaunchKernelGGL(traceback_t_kernel, dim3((a.n_launch + WSX_GEN_RPW - 1) / WSX_GEN_RPW), dim3(64), shmem, s, a, nwp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return wsx_launch_expand_trace(a, s);
}

