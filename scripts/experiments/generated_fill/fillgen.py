"""Generated DP fill for automata of up to 64 states: a read in FOUR lanes, straight-line code per automaton.

The register-resident fill of csrc/dtw_kernels.hip keeps a state per lane and hands predecessor values over through LDS;
every state pays for as many candidates as the busiest one (10 vector instructions per row of 64 states with two
candidates).  Almost all states of a k-mer automaton are chain states with ONE predecessor.  With the states of a read laid
along the registers of four neighbouring lanes (16 reads per wavefront) a chain state's predecessor is the register next
door -- no exchange at all -- and the few other edges (loop entries, merges) become extra add / compare / min groups under a
constant exec mask; the value crossing a lane boundary travels by two DPP moves.  A chain state then costs 7 vector
instructions per row instead of 10, at the price of code that is specific to the automaton: this module writes that code
(`generate`), compiles it at run time (`compile_source`: hiprtc in process, else `hipcc --genco`; cached by the hash of the
source) and hands the code object to the library (`wsx_caller_set_generated_fill`), which launches it in place of
`dtw_fill_fast` for the reads of that automaton; `traceback_t_kernel` walks the back-pointer words it writes.  The
arithmetic per cell is the reference's, operation for operation (caller.py:198-245): the same
left-to-right dwell sums, stay first, predecessors in `incoming` order with strict `<`, the forced first rows, the corner
cut, `back = m - 1` on masked samples.  If generation or compilation fails the handle keeps `dtw_fill_fast`.

Measured first in round 2 as an experiment (scripts/exp_transposed_gen.py, profiles/r02_transposed_layout.log).
"""
import ctypes as C
import hashlib
import os
import subprocess
import tempfile
from collections import defaultdict
from typing import Dict, List, Optional, Tuple

import numpy as np

G = 4            # lanes per read
RPW = 64 // G    # reads per wavefront
M = 4            # min_values_per_state this code is written for
MAX_F = 4        # candidates per state the traceback table holds
ABI = 1          # layout of GenFillArgs / the tables below; the library checks it


def _linearize(S: int, preds: List[List[int]]) -> List[int]:
    """States in an order where a state's first predecessor is the element before it as often as possible."""
    fp = [p[0] if len(p) else -1 for p in preds]
    kids = defaultdict(list)
    for j in range(S):
        if fp[j] >= 0:
            kids[fp[j]].append(j)
    height: Dict[int, int] = {}
    for j in sorted(range(S), reverse=True):  # iterative heights (a first-predecessor cycle is cut where it is met)
        stack = [(j, iter(kids[j]))]
        on_path = {j}
        while stack:
            x, it = stack[-1]
            nxt = next(it, None)
            if nxt is None:
                height[x] = 1 + max([height.get(c, 0) for c in kids[x]] or [0])
                on_path.discard(x)
                stack.pop()
            elif nxt not in height and nxt not in on_path:
                on_path.add(nxt)
                stack.append((nxt, iter(kids[nxt])))
    order: List[int] = []
    seen = set()

    def emit(j):
        stack = [j]
        while stack:
            x = stack.pop()
            if x in seen:
                continue
            seen.add(x)
            order.append(x)
            for c in sorted(kids[x], key=lambda c: -height.get(c, 0)):  # the longest tail is popped last... first pushed
                stack.append(c)
    for j in range(S):
        if fp[j] < 0 or j == 0:
            emit(j)
    for j in range(S):
        if j not in seen:
            emit(j)
    return order


def _quad_perm(d: int) -> int:
    sel = [min(max(i - d, 0), 3) for i in range(4)]
    return sel[0] | sel[1] << 2 | sel[2] << 4 | sel[3] << 6


class Generated:
    """Source text + what the library needs beside the code object."""

    def __init__(self, source: str, words_per_row: int, n_per_lane: int, state_at: np.ndarray, tb_n: np.ndarray,
                 tb_word: np.ndarray, tb_pred: np.ndarray, end_pos: int, valu_per_wave_row: int):
        self.source, self.words_per_row, self.n_per_lane = source, words_per_row, n_per_lane
        self.state_at, self.tb_n, self.tb_word, self.tb_pred, self.end_pos = state_at, tb_n, tb_word, tb_pred, end_pos
        self.valu_per_wave_row = valu_per_wave_row
        self.key = hashlib.sha256(source.encode()).hexdigest()[:24]


def supported(table, min_values_per_state: int) -> bool:
    return (min_values_per_state == M and M < table.n_states <= 64 and table.max_fanin <= MAX_F)


def _opts() -> Dict[str, int]:
    """Generator knobs (experiments; part of the source and therefore of the cache key): WARPSTR_FILLGEN_OPTS="sb=1,wpe=2"."""
    o = {'sb': 1, 'wpe': 2, 'prio': 3}
    for kv in filter(None, os.environ.get('WARPSTR_FILLGEN_OPTS', '').split(',')):
        k, v = kv.split('=')
        o[k] = int(v)
    return o


def generate(table, flank_length: int) -> Generated:
    """HIP source of the two fill kernels (wsx_fill_t_u: unmasked pass, wsx_fill_t_m: a bad-repeat mask per read) for one
    automaton (warpstr_amd.automata.AutomatonTable) and the tables of its back-pointer layout."""
    S = int(table.n_states)
    opt = _opts()
    values = np.asarray(table.value, np.float64)
    pp, pi = np.asarray(table.pred_ptr), np.asarray(table.pred_idx)
    preds = [[int(x) for x in pi[pp[j]:pp[j + 1]]] for j in range(S)]
    n = (S + G - 1) // G
    L = _linearize(S, preds)
    pos = {j: (p // n, p % n) for p, j in enumerate(L)}
    state_at = [[-1] * n for _ in range(G)]
    for j, (q, k) in pos.items():
        state_at[q][k] = j
    boundary = int(flank_length) - 10
    after_repeat = int(table.seq_idx[S - 1]) - boundary
    cut_state = [[(state_at[q][k] >= 0 and int(table.seq_idx[state_at[q][k]]) < after_repeat) for k in range(n)] for q in range(G)]

    def lanes_mask(qs):
        m = 0
        for g in range(RPW):
            for q in qs:
                m |= 1 << (g * G + q)
        return m
    ALL = (1 << 64) - 1
    groups = [defaultdict(lambda: defaultdict(set)) for _ in range(n)]  # groups[k][f][(kp, d)] = lanes q
    for j in range(S):
        q, k = pos[j]
        for f, p in enumerate(preds[j]):
            qp, kp = pos[p]
            groups[k][f][(kp, q - qp)].add(q)
    transports = sorted({key for k in range(n) for f in groups[k] for key in groups[k][f] if key[1] != 0})
    hoisted = []  # same lane, source above the consumer (updated earlier in the descending sweep): candidate taken before it
    n_valu = 0
    code: List[str] = []
    w = code.append
    words = []  # (k, f, kp, d, lanes)
    tname = lambda kp, d: f't_{kp}_{d if d > 0 else "m%d" % -d}'
    for (kp, d) in transports:
        ctrl = _quad_perm(d)
        w(f'        const double {tname(kp, d)} = shift<{ctrl}>(c3_{kp});')
        n_valu += 2
    for k in range(n):
        for f in sorted(groups[k]):
            for (kp, d) in sorted(groups[k][f]):
                if d == 0 and kp > k:
                    hoisted.append((k, f, kp))
    pre_a = sorted({k for (k, f, kp) in hoisted})
    for k in pre_a:
        w(f'        const double a_{k} = s - v_{k};')
        n_valu += 1
    for (k, f, kp) in hoisted:
        w(f'        const double h_{k}_{f}_{kp} = add_abs(c3_{kp}, a_{k});')
        n_valu += 1
    for k in range(n - 1, -1, -1):
        if k not in pre_a:
            w(f'        const double a_{k} = s - v_{k};')
            n_valu += 1
        w(f'        const double stay_{k} = add_abs(D_{k}, a_{k});')
        w(f'        double best_{k} = stay_{k};')
        n_valu += 1
        for f in sorted(groups[k]):
            for (kp, d) in sorted(groups[k][f]):
                qs = groups[k][f][(kp, d)]
                mask = lanes_mask(qs)
                wi = len(words)
                words.append((k, f, kp, d, sorted(qs)))
                if d == 0 and kp > k:
                    w(f'        u64 w{wi} = grp_cm(best_{k}, h_{k}_{f}_{kp}, 0x{mask:016x}ull);')
                    n_valu += 2
                else:
                    src = f'c3_{kp}' if d == 0 else tname(kp, d)
                    if mask == ALL:
                        w(f'        u64 w{wi} = grp_all(best_{k}, {src}, a_{k});')
                    else:
                        w(f'        u64 w{wi} = grp(best_{k}, {src}, a_{k}, 0x{mask:016x}ull);')
                    n_valu += 3
                w(f'        if (CUT) w{wi} &= ~cutm_{k};')
        w(f'        if (CUT) best_{k} = cutf_{k} ? kInf : best_{k};')
        w(f'        c3_{k} = add_abs(c2_{k}, a_{k}); if (MASKED) c3_{k} = add_abs_masked(c3_{k}, c1_{k}, a_{k}, mnext); '
          f'c2_{k} = add_abs(c1_{k}, a_{k}); c1_{k} = stay_{k}; D_{k} = best_{k};')
        n_valu += 2
    NW = len(words)
    NWP = max((NW + 1) & ~1, 2)
    if NWP > 32:
        raise ValueError(f'{NW} back-pointer words per row: more than the traceback stages through LDS (32)')
    for i in range(0, NW, 2):
        if i + 1 < NW:
            w(f'        store2<{i * 8}>(w{i}, w{i + 1}, rowp);')
        else:
            w(f'        store1<{i * 8}>(w{i}, rowp);')
    body = '\n'.join(code)
    regs = ' '.join(f'double D_{k}, c1_{k}, c2_{k}, c3_{k}, v_{k};' for k in range(n))
    vtab = [[float(values[state_at[q][k]]) if state_at[q][k] >= 0 else 0.0 for k in range(n)] for q in range(G)]
    init = '\n'.join(
        f'    v_{k} = vtab[q][{k}]; {{ const int j = jtab[q][{k}]; double d0 = kInf; if (j == 0) d0 = start; '
        f'else if (j > 0 && j <= M) d0 = start + fabs(smp(j) - v0); D_{k} = d0; }} c1_{k} = kInf; c2_{k} = kInf; c3_{k} = kInf;'
        for k in range(n))
    adv = '\n'.join(
        f'        {{ const double a = s - v_{k}; c3_{k} = add_abs(c2_{k}, a); if (MASKED) c3_{k} = add_abs_masked(c3_{k}, c1_{k}, a, mnext); '
        f'c2_{k} = add_abs(c1_{k}, a); c1_{k} = add_abs(D_{k}, a); D_{k} = kInf; }}' for k in range(n))
    cut_decl = '\n'.join(
        f'        const bool cutf_{k} = CUT && cuttab[q][{k}] && cut_now; const u64 cutm_{k} = CUT ? __ballot(cutf_{k}) : 0ull;'
        for k in range(n))
    last_out = '\n'.join(
        f'            {{ const int j = jtab[q][{k}]; if (j >= 0) a.last_row[(size_t)lr * a.last_row_stride + j] = D_{k}; }}' for k in range(n))
    eq, ek = pos[int(table.endstate)]
    end_sel = f'D_{ek}'
    sbmacro = '__builtin_amdgcn_sched_barrier(0);' if opt['sb'] else ''
    wpe = f"__attribute__((amdgpu_waves_per_eu({opt['wpe']})))" if opt['wpe'] else ''
    brace = lambda rows, fmt: ', '.join('{' + ', '.join(fmt(x) for x in row) + '}' for row in rows)
    src = f'''// generated by warpstr_amd/fillgen.py (ABI {ABI}): S={S}, {G} lanes per read, {n} states per lane, {NW} back-pointer words per
// wave-row, {n_valu} vector instructions per wave-row by construction ({n_valu * G / 64:.2f} per read-row)
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif
namespace {{
typedef unsigned long long u64;
typedef long long i64;
typedef unsigned int u32;
typedef int i32;
struct No {{ static constexpr bool value = false; }};
struct Yes {{ static constexpr bool value = true; }};
constexpr double kInf = __builtin_huge_val();
constexpr int G = {G}, N = {n}, RPW = {RPW}, NW = {NW}, NWP = {NWP}, M = {M};
__device__ const double vtab[G][N] = {{{brace(vtab, repr)}}};
__device__ const int jtab[G][N] = {{{brace(state_at, str)}}};
__device__ const bool cuttab[G][N] = {{{brace(cut_state, lambda b: 'true' if b else 'false')}}};
struct GenFillArgs {{ // = wsx_api.hip: GenFillArgs (ABI {ABI})
    const double *signal;      // the chunk's samples (index loff + i)
    const i64 *offsets;    // global offsets[], by global read id
    const i32 *order;      // read ids of this launch, longest first
    const i64 *bp_off;     // per read of the chunk (index lr): start of its WAVE's back-pointer rows, in 64-bit words
    u64 *bp;
    const u32 *maskbits;  // packed sample masks (word loff/32 + lr + w), or null
    double *end_cost;          // per read, or null
    double *last_row;          // per read, stride last_row_stride, or null
    i32 *status;           // per read
    i64 base_off;
    i32 n_launch, first_read, last_row_stride, check_status;
    i32 boundary;          // flank_length - 10 of the automaton
    i32 pad;
}};
__device__ __forceinline__ double add_abs(double x, double a) {{ double r; asm("v_add_f64 %0, %1, |%2|" : "=v"(r) : "v"(x), "v"(a)); return r; }}
__device__ __forceinline__ double min_f64(double a, double b) {{ double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }}
// keep, except in the lanes of `mask`: x + |a|  (the per-read choice of the export under the bad-repeat mask)
__device__ __forceinline__ double add_abs_masked(double keep, double x, double a, u64 mask)
{{
    asm("s_mov_b64 exec, %3\\n\\tv_add_f64 %0, %1, |%2|\\n\\ts_mov_b64 exec, -1" : "+v"(keep) : "v"(x), "v"(a), "s"(mask));
    return keep;
}}
template <int CTRL> __device__ __forceinline__ double shift(double x)
{{
    long long b = __double_as_longlong(x);
    int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}}
__device__ __forceinline__ u64 grp_all(double &best, double src, double a)
{{
    const double c = add_abs(src, a);
    u64 lt;
    asm("v_cmp_lt_f64 %0, %1, %2" : "=s"(lt) : "v"(c), "v"(best));
    best = min_f64(best, c);
    return lt;
}}
__device__ __forceinline__ u64 grp(double &best, double src, double a, u64 mask)
{{
    double t; u64 lt;
    asm("s_mov_b64 exec, %4\\n\\tv_add_f64 %0, %3, |%5|\\n\\tv_cmp_lt_f64 %1, %0, %2\\n\\tv_min_f64 %2, %2, %0\\n\\ts_mov_b64 exec, -1"
        : "=&v"(t), "=&s"(lt), "+v"(best) : "v"(src), "s"(mask), "v"(a));
    return lt;
}}
__device__ __forceinline__ u64 grp_cm(double &best, double c, u64 mask)
{{
    u64 lt;
    asm("s_mov_b64 exec, %3\\n\\tv_cmp_lt_f64 %0, %2, %1\\n\\tv_min_f64 %1, %1, %2\\n\\ts_mov_b64 exec, -1"
        : "=&s"(lt), "+v"(best) : "v"(c), "s"(mask));
    return lt;
}}
template <int OFF> __device__ __forceinline__ void store1(u64 m, u64 *p) {{ asm volatile("s_store_dwordx2 %0, %1, %2" ::"s"(m), "s"(p), "n"(OFF) : "memory"); }}
template <int OFF> __device__ __forceinline__ void store2(u64 m0, u64 m1, u64 *p)
{{
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = {{(u32)m0, (u32)(m0 >> 32), (u32)m1, (u32)(m1 >> 32)}};
    asm volatile("s_store_dwordx4 %0, %1, %2" ::"s"(v), "s"(p), "n"(OFF) : "memory");
}}
__device__ __forceinline__ u64 rfl64(u64 x) // a wave-uniform 64-bit value, in scalar registers (the scalar stores need their base there)
{{
    return ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(x >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)x);
}}
__device__ __forceinline__ int wave_min(int x)
{{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {{ const int y = __shfl_xor(x, d, 64); x = y < x ? y : x; }}
    return x;
}}
__device__ __forceinline__ int wave_max(int x)
{{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {{ const int y = __shfl_xor(x, d, 64); x = y > x ? y : x; }}
    return x;
}}

template <bool MASKED>
__device__ __forceinline__ void fill_body(const GenFillArgs &a)
{{
    __builtin_amdgcn_s_setprio({opt['prio']});
    // Who this quad is: its read's id, length, place in the chunk.  Looked up again after the long loop (`salt` keeps the two
    // look-ups apart): the DP state of 16 positions takes nearly all of the 256 registers two waves per SIMD leave, and whatever
    // is live across the loop is spilled inside it.
    struct Geo {{
        int lr, T, Tr, q;
        bool live;
        const double *sp;
        const u32 *mw;
    }};
    const int Tmax = __builtin_amdgcn_readfirstlane((int)(a.offsets[a.order[blockIdx.x * RPW] + 1] - a.offsets[a.order[blockIdx.x * RPW]]));
    if (Tmax <= M) {{ // every read of the wavefront is too short (the launch order is longest first): WSX_READ_SHAPE
        const int slot = blockIdx.x * RPW + (int)threadIdx.x / G;
        if (slot < a.n_launch && threadIdx.x % G == 0 && !a.check_status) {{
            const int lr = a.order[slot] - a.first_read;
            a.status[lr] = 1;
            if (a.end_cost) a.end_cost[lr] = kInf;
        }}
        return;
    }}
    auto geo = [&](int salt) -> Geo {{
        int lane = threadIdx.x;
        asm volatile("; geo %1" : "+v"(lane) : "n"(0) : );
        (void)salt;
        Geo me;
        const int g = lane / G;
        me.q = lane % G;
        int slot = blockIdx.x * RPW + g;
        me.live = slot < a.n_launch;
        const int r0 = a.order[blockIdx.x * RPW];
        const int r = me.live ? a.order[slot] : r0; // an idle quad shadows the wave's first read (it writes nothing)
        me.lr = r - a.first_read;
        const long long off = a.offsets[r] - a.base_off;
        me.T = (int)(a.offsets[r + 1] - a.offsets[r]);
        if (me.live && a.check_status && a.status[me.lr] != 0) me.live = false; // second pass: a read that failed before is left alone
        if (me.T <= M) me.live = false;                                          // too short: handled below, runs along as a shadow
        // a quad without a read of its own (idle, failed, too short) runs along on the first read's samples
        const long long off0 = a.offsets[r0] - a.base_off;
        me.sp = a.signal + (me.live ? off : off0);
        me.Tr = me.live ? me.T : Tmax;
        me.mw = a.maskbits + ((me.live ? off : off0) / 32 + (me.live ? me.lr : r0 - a.first_read)); // (only dereferenced when there is a mask)
        return me;
    }};
    const bool has_mask = MASKED && a.maskbits != nullptr; // (wave-uniform: a kernel argument)
    u64 *wbp = (u64 *)rfl64((u64)(a.bp + a.bp_off[a.order[blockIdx.x * RPW] - a.first_read]));
    int cut_lo, cut_hi, Tmin;
    const double *sp;
    const u32 *mw;
    int Tr, q;
    {regs}
    {{
        const Geo me = geo(0);
        sp = me.sp;
        mw = me.mw;
        Tr = me.Tr;
        q = me.q;
        if (me.T <= M && q == 0 && !a.check_status) {{ // WSX_READ_SHAPE (upstream: IndexError, caller.py:206-208)
            const int slot = blockIdx.x * RPW + (int)threadIdx.x / G;
            if (slot < a.n_launch) {{
                a.status[me.lr] = 1;
                if (a.end_cost) a.end_cost[me.lr] = kInf;
            }}
        }}
        // corner cut (caller.py:211-224): from row cut_from on, the states before the repeat's end region are removed; forcing it
        // for M rows is enough (a cut state's predecessors are cut states: csrc/dtw_kernels.hip).  Reads of a wave differ in
        // length, so the rows that force it run from the earliest start to the latest start + M, each quad applying its own.
        const long long b6 = 6ll * a.boundary;
        const long long cf = b6 > (long long)Tr - b6 + 1 ? b6 : (long long)Tr - b6 + 1;
        const int cut_from = cf < M ? M : (cf > Tr ? Tr : (int)cf);
        cut_lo = __builtin_amdgcn_readfirstlane(wave_min(cut_from));
        const int cut_hi0 = __builtin_amdgcn_readfirstlane(wave_max(cut_from)) + M;
        cut_hi = cut_hi0 < Tmax ? cut_hi0 : Tmax;
        Tmin = __builtin_amdgcn_readfirstlane(wave_min(Tr));
    }}
    auto smp = [&](int i) -> double {{ return sp[i < Tr ? i : Tr - 1]; }};
    const int nmw = (Tr + 31) / 32;
    const double v0 = vtab[0][0];
    const double start = fabs(smp(0) - v0);
{init}
    auto mask_word = [&](int w0) -> u32 {{ return (has_mask && w0 < nmw) ? mw[w0] : 0u; }};
    auto mask_window = [&](int b) -> u64 {{ // per lane: bits of its read's samples b+1 .. (two words: the window may cross one)
        const int w0 = (b + 1) >> 5;
        const u64 lo = mask_word(w0), hi = mask_word(w0 + 1);
        return (lo | (hi << 32)) >> ((b + 1) & 31);
    }};
    auto mrow = [&](int rr) -> u64 {{ // the quads whose read has sample rr masked
        const u32 word = rr < Tr ? mask_word(rr >> 5) : 0u;
        return __ballot((word >> (rr & 31)) & 1u);
    }};
    for (int i = 1; i < M; i++) {{ // rows 1 .. M-1: D stays +inf, only the dwell sums advance (caller.py:217 starts at i = m)
        const double s = smp(i);
        const u64 mnext = mrow(i + 1);
{adv}
    }}
    auto row = [&](auto cut_tag, int cut_from, int i, double s, u64 *rowp, u64 mnext) __attribute__((always_inline)) {{
        constexpr bool CUT = decltype(cut_tag)::value;
        const bool cut_now = CUT && i >= cut_from;
{cut_decl}
{body}
    }};
    typedef double d2 __attribute__((ext_vector_type(2)));
    // two samples per load; only rows below the shortest read's last one are taken this way (the clamp keeps the prefetch of
    // the round after the last inside the read)
    auto pair = [&](int rr) -> d2 {{ const int c = rr + 2 <= Tr ? rr : Tr - 2; return *(const d2 *)(sp + c); }};
    // plain rows [lo, hi): eight per round, the samples arriving two at a time with two loads in flight
    auto plain = [&](int lo, int hi) __attribute__((always_inline)) {{
        int i = lo;
        for (; i < hi && (i & 7); i++) row(No{{}}, 0, i, smp(i), wbp + (size_t)i * NWP, mrow(i + 1));
        if (i + 8 <= hi) {{
            d2 p0 = pair(i), p1 = pair(i + 2);
            for (; i + 8 <= hi; i += 8) {{
                const u64 win = mask_window(i);
                u64 mm[8];
#pragma unroll
                for (int u = 0; u < 8; u++) mm[u] = MASKED ? __ballot((win >> u) & 1ull) : 0ull;
                u64 *rp = wbp + (size_t)i * NWP;
#define WSX_SB {sbmacro} // rows are not interleaved: 16 independent positions are parallelism enough, and values of two rows in flight at once are what the register file has no room for
                row(No{{}}, 0, i, p0.x, rp, mm[0]); WSX_SB row(No{{}}, 0, i + 1, p0.y, rp + NWP, mm[1]); p0 = pair(i + 4); WSX_SB
                row(No{{}}, 0, i + 2, p1.x, rp + 2 * NWP, mm[2]); WSX_SB row(No{{}}, 0, i + 3, p1.y, rp + 3 * NWP, mm[3]); p1 = pair(i + 6); WSX_SB
                row(No{{}}, 0, i + 4, p0.x, rp + 4 * NWP, mm[4]); WSX_SB row(No{{}}, 0, i + 5, p0.y, rp + 5 * NWP, mm[5]); p0 = pair(i + 8); WSX_SB
                row(No{{}}, 0, i + 6, p1.x, rp + 6 * NWP, mm[6]); WSX_SB row(No{{}}, 0, i + 7, p1.y, rp + 7 * NWP, mm[7]); p1 = pair(i + 10); WSX_SB
#undef WSX_SB
            }}
        }}
        for (; i < hi; i++) row(No{{}}, 0, i, smp(i), wbp + (size_t)i * NWP, mrow(i + 1));
    }};
    // rows [lo, hi) one at a time, with everything a row may need: the forced cut, and the capture of a read's last row
    auto careful = [&](int lo, int hi, int salt) __attribute__((always_inline)) {{
        if (lo >= hi) return;
        const Geo me = geo(salt);
        const long long b6 = 6ll * a.boundary;
        const long long cf = b6 > (long long)me.Tr - b6 + 1 ? b6 : (long long)me.Tr - b6 + 1;
        const int cut_from = cf < M ? M : (cf > me.Tr ? me.Tr : (int)cf);
        for (int i = lo; i < hi; i++) {{
            if (i >= cut_lo && i < cut_hi) row(Yes{{}}, cut_from, i, smp(i), wbp + (size_t)i * NWP, mrow(i + 1));
            else row(No{{}}, 0, i, smp(i), wbp + (size_t)i * NWP, mrow(i + 1));
            if (me.live && me.T == i + 1) {{ // row T-1 of this quad's read has just been computed
                if (a.end_cost && jtab[me.q][{ek}] == {int(table.endstate)}) a.end_cost[me.lr] = {end_sel};
                if (a.last_row) {{
                    const int lr = me.lr, q = me.q;
{last_out}
                }}
                if (me.q == 0 && !a.check_status) a.status[me.lr] = 0;
            }}
        }}
    }};
    // [M, Tmax) = plain rows, except the rows that force a cut and the rows from the shortest read's last one on
    const int tail = Tmin - 1 > M ? Tmin - 1 : M;          // the first row after which some read ends
    const int c_lo = cut_lo < tail ? cut_lo : tail, c_hi = cut_hi < tail ? cut_hi : tail;
    plain(M, c_lo);
    careful(c_lo, c_hi, 1);
    plain(c_hi, tail);
    careful(tail, Tmax, 2);
    asm volatile("s_dcache_wb" ::: "memory"); // scalar stores sit in the scalar data cache until written back
}}
}} // namespace

extern "C" __global__ __launch_bounds__(64) {wpe} void wsx_fill_t_u(GenFillArgs a) {{ fill_body<false>(a); }}
extern "C" __global__ __launch_bounds__(64) {wpe} void wsx_fill_t_m(GenFillArgs a) {{ fill_body<true>(a); }}
'''
    # tables for the traceback: per position p = q * n + k the candidates in `incoming` order: word of the row, predecessor position
    P = G * n
    tb_n = np.zeros(P, np.uint8)
    tb_word = np.zeros((P, MAX_F), np.uint16)
    tb_pred = np.zeros((P, MAX_F), np.uint16)
    for wi, (k, f, kp, d, qs) in enumerate(words):
        for q in qs:
            p = q * n + k
            tb_word[p, f] = wi
            tb_pred[p, f] = (q - d) * n + kp
            tb_n[p] = max(tb_n[p], f + 1)
    sa = np.full(P, 0xFFFF, np.uint16)
    for j, (q, k) in pos.items():
        sa[q * n + k] = j
    return Generated(src, NWP, n, sa, tb_n, tb_word.reshape(-1).copy(), tb_pred.reshape(-1).copy(), eq * n + ek, n_valu)


# ---- compilation ------------------------------------------------------------------------------------------------------------
def cache_dir() -> str:
    """Where code objects are kept: WARPSTR_CACHE_DIR, else ~/.cache/warpstr_amd/fillgen, else (home not writable) a per-user
    directory under the system's temporary directory."""
    choices = [os.environ.get('WARPSTR_CACHE_DIR'), os.path.join(os.path.expanduser('~'), '.cache', 'warpstr_amd', 'fillgen'),
               os.path.join(tempfile.gettempdir(), f'warpstr_amd_fillgen_{os.getuid()}')]
    for d in filter(None, choices):
        try:
            os.makedirs(d, exist_ok=True)
            if os.access(d, os.W_OK):
                return d
        except OSError:
            continue
    return tempfile.mkdtemp(prefix='warpstr_amd_fillgen_')


_OPTS = ['--offload-arch=gfx950', '-O3', '-ffp-contract=off', '-std=c++17']


def _hiprtc_lib():
    import sys
    names = []
    torch = sys.modules.get('torch')
    if torch is not None:  # the runtime torch brought: the one this process already uses
        names.append(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libhiprtc.so'))
    names += ['libhiprtc.so', '/opt/rocm/lib/libhiprtc.so']
    for nm in names:
        try:
            return C.CDLL(nm)
        except OSError:
            continue
    return None


def _compile_hiprtc(source: str) -> Optional[bytes]:
    rtc = _hiprtc_lib()
    if rtc is None:
        return None
    prog = C.c_void_p()
    if rtc.hiprtcCreateProgram(C.byref(prog), source.encode(), b'wsx_fill_t.hip', 0, None, None) != 0:
        return None
    opts = (C.c_char_p * len(_OPTS))(*[o.encode() for o in _OPTS])
    rc = rtc.hiprtcCompileProgram(prog, len(_OPTS), opts)
    try:
        if rc != 0:
            return None
        n = C.c_size_t()
        rtc.hiprtcGetCodeSize(prog, C.byref(n))
        code = C.create_string_buffer(n.value)
        rtc.hiprtcGetCode(prog, code)
        return code.raw
    finally:
        rtc.hiprtcDestroyProgram(C.byref(prog))


def _compile_hipcc(source: str) -> Optional[bytes]:
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    if not os.path.exists(hipcc):
        return None
    with tempfile.TemporaryDirectory() as d:
        src, out = os.path.join(d, 'wsx_fill_t.hip'), os.path.join(d, 'wsx_fill_t.hsaco')
        with open(src, 'w') as f:
            f.write(source)
        rc = subprocess.run([hipcc, '--genco'] + _OPTS + [src, '-o', out], capture_output=True)
        if rc.returncode != 0 or not os.path.exists(out):
            if os.environ.get('WARPSTR_FILLGEN_VERBOSE'):
                print(rc.stderr.decode()[-4000:])
            return None
        with open(out, 'rb') as f:
            return f.read()


def compile_source(gen: Generated, compile_missing: bool = True) -> Tuple[Optional[bytes], str]:
    """Code object of a generated source: from the cache, else (compile_missing) hiprtc in process, else `hipcc --genco`;
    (None, why) if none of them delivers.  Cached by the hash of the source (the automaton's levels and edges are part of it)."""
    path = os.path.join(cache_dir(), gen.key + '.hsaco')
    if os.path.exists(path) and os.path.getsize(path) > 0:
        with open(path, 'rb') as f:
            return f.read(), 'cache'
    if not compile_missing:
        return None, 'not in the cache'
    how = 'hiprtc'
    code = None if os.environ.get('WARPSTR_FILLGEN_HIPCC') else _compile_hiprtc(gen.source)
    if code is None:
        how = 'hipcc'
        code = _compile_hipcc(gen.source)
    if code is None:
        return None, 'no compiler (hiprtc and hipcc both failed)'
    tmp = path + f'.{os.getpid()}.tmp'
    with open(tmp, 'wb') as f:
        f.write(code)
    os.replace(tmp, path)
    return code, how
