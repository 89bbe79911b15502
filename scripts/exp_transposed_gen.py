"""Experiment: the DP fill with a read in G lanes (n = ceil(S/G) states per lane, all of a lane's state in registers)
instead of a state per lane -- no LDS exchange; the code is generated for ONE automaton (straight-line, unrolled over the
positions).  Writes a .hip file with the kernel and a small C driver (run_fill_t).

    python scripts/exp_transposed_gen.py headline 4 build/exp/fill_t.hip [--masked]   (--masked: the second pass, a bad-repeat mask per read)
"""
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def linearize(S, preds):
    """States in an order where a state's first predecessor is the previous element as often as possible."""
    fp = [p[0] if len(p) else -1 for p in preds]
    kids = defaultdict(list)
    for j in range(S):
        if fp[j] >= 0:
            kids[fp[j]].append(j)
    height = {}

    def h(j, seen=()):
        if j in height:
            return height[j]
        if j in seen:
            return 0
        height[j] = 1 + max([h(c, seen + (j,)) for c in kids[j]] or [0])
        return height[j]
    order, seen = [], set()

    def emit(j):
        stack = [j]
        while stack:
            x = stack.pop()
            if x in seen:
                continue
            seen.add(x)
            order.append(x)
            # shortest subtree first (popped first = pushed last)
            for c in sorted(kids[x], key=lambda c: -h(c)):
                stack.append(c)
    for j in range(S):
        if fp[j] < 0 or j == 0:
            emit(j)
    for j in range(S):
        if j not in seen:
            emit(j)
    return order


def quad_perm(d):
    sel = [min(max(i - d, 0), 3) for i in range(4)]
    return sel[0] | sel[1] << 2 | sel[2] << 4 | sel[3] << 6


def dpp_ctrl(G, d):
    if G == 4:
        return quad_perm(d)
    return (0x110 + d) if d > 0 else (0x100 - d)


def generate(values, preds, endstate, G, out, M=4, masked=False):
    S = len(values)
    n = (S + G - 1) // G
    L = linearize(S, preds)
    pos = {}
    for p, j in enumerate(L):
        pos[j] = (p // n, p % n)
    state_at = [[-1] * n for _ in range(G)]
    for j, (q, k) in pos.items():
        state_at[q][k] = j
    RPW = 64 // G

    def lanes_mask(qs):
        m = 0
        for g in range(RPW):
            for q in qs:
                m |= 1 << (g * G + q)
        return m
    ALL = (1 << 64) - 1
    # groups[k][f] : dict (kp, d) -> set of lanes q
    groups = [defaultdict(lambda: defaultdict(set)) for _ in range(n)]
    for j in range(S):
        q, k = pos[j]
        for f, p in enumerate(preds[j]):
            qp, kp = pos[p]
            groups[k][f][(kp, q - qp)].add(q)
    transports = sorted({key for k in range(n) for f in groups[k] for key in groups[k][f] if key[1] != 0})
    hoisted = []  # (k, f, kp): same lane, source position above the consumer's (updated earlier in the descending sweep)
    n_valu = 0
    code = []
    w = code.append
    words = []
    w('    // transports: c3 of position kp from the lane d to the left')
    for (kp, d) in transports:
        w(f'    const double t_{kp}_{d if d > 0 else "m%d" % -d} = shift<{dpp_ctrl(G, d)}>(c3_{kp});')
        n_valu += 2
    for k in range(n):
        for f in sorted(groups[k]):
            for (kp, d) in sorted(groups[k][f]):
                if d == 0 and kp > k:
                    hoisted.append((k, f, kp))
    pre_a = sorted({k for (k, f, kp) in hoisted})
    for k in pre_a:
        w(f'    const double a_{k} = s - v_{k};')
        n_valu += 1
    for (k, f, kp) in hoisted:
        w(f'    const double h_{k}_{f}_{kp} = add_abs(c3_{kp}, a_{k});')
        n_valu += 1
    n_groups = 0
    for k in range(n - 1, -1, -1):
        if k not in pre_a:
            w(f'    const double a_{k} = s - v_{k};')
            n_valu += 1
        w(f'    const double stay_{k} = add_abs(D_{k}, a_{k});')
        w(f'    double best_{k} = stay_{k};')
        n_valu += 1
        for f in sorted(groups[k]):
            for (kp, d) in sorted(groups[k][f]):
                qs = groups[k][f][(kp, d)]
                mask = lanes_mask(qs)
                wi = len(words)
                words.append((k, f, kp, d, sorted(qs)))
                tname = f't_{kp}_{d if d > 0 else "m%d" % -d}'
                if d == 0 and kp > k:
                    w(f'    const uint64_t w{wi} = grp_cm(best_{k}, h_{k}_{f}_{kp}, 0x{mask:016x}ull);')
                    n_valu += 2
                else:
                    src = f'c3_{kp}' if d == 0 else tname
                    if mask == ALL:
                        w(f'    const uint64_t w{wi} = grp_all(best_{k}, {src}, a_{k});')
                    else:
                        w(f'    const uint64_t w{wi} = grp(best_{k}, {src}, a_{k}, 0x{mask:016x}ull);')
                    n_valu += 3
                n_groups += 1
        if masked:  # c3 is only ever read as the export of the NEXT row: (that row masked for this read ? c1 : c2) + a
            w(f'    c3_{k} = add_abs(c2_{k}, a_{k}); c3_{k} = add_abs_masked(c3_{k}, c1_{k}, a_{k}, mnext); c2_{k} = add_abs(c1_{k}, a_{k}); c1_{k} = stay_{k}; D_{k} = best_{k};')
            n_valu += 3
        else:
            w(f'    c3_{k} = add_abs(c2_{k}, a_{k}); c2_{k} = add_abs(c1_{k}, a_{k}); c1_{k} = stay_{k}; D_{k} = best_{k};')
            n_valu += 2
    NW = len(words)
    NWP = (NW + 1) & ~1
    for i in range(0, NW, 2):
        if i + 1 < NW:
            w(f'    store2<{i * 8}>(w{i}, w{i + 1}, rowp);')
        else:
            w(f'    store1<{i * 8}>(w{i}, rowp);')
    body = '\n'.join(code)
    regs = ' '.join(f'double D_{k}, c1_{k}, c2_{k}, c3_{k}, v_{k};' for k in range(n))
    vtab = []
    for q in range(G):
        vtab.append([float(values[state_at[q][k]]) if state_at[q][k] >= 0 else 0.0 for k in range(n)])
    jtab = [[state_at[q][k] for k in range(n)] for q in range(G)]
    eq, ek = pos[endstate]
    src = f'''// generated by scripts/exp_transposed_gen.py: S={S} G={G} n={n} words/row={NW} VALU/wave-row={n_valu} ({n_valu * G / 64:.2f} per read-row)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
namespace {{
constexpr double kInf = __builtin_huge_val();
constexpr int G = {G}, N = {n}, RPW = {RPW}, NW = {NW}, NWP = {NWP}, M = {M};
constexpr bool MASKED = {'true' if masked else 'false'};
__device__ const double vtab[G][N] = {{{', '.join('{' + ', '.join(repr(x) for x in row) + '}' for row in vtab)}}};
__device__ const int jtab[G][N] = {{{', '.join('{' + ', '.join(str(x) for x in row) + '}' for row in jtab)}}};
__device__ __forceinline__ double add_abs(double x, double a) {{ double r; asm("v_add_f64 %0, %1, |%2|" : "=v"(r) : "v"(x), "v"(a)); return r; }}
__device__ __forceinline__ double min_f64(double a, double b) {{ double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }}
// keep, except in the lanes of `mask`: x + |a|  (the per-read choice of the export under the bad-repeat mask)
__device__ __forceinline__ double add_abs_masked(double keep, double x, double a, uint64_t mask)
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
__device__ __forceinline__ uint64_t grp_all(double &best, double src, double a)
{{
    const double c = add_abs(src, a);
    const uint64_t lt = __builtin_amdgcn_fcmp(c, best, 4);
    best = min_f64(best, c);
    return lt;
}}
__device__ __forceinline__ uint64_t grp(double &best, double src, double a, uint64_t mask)
{{
    double t; uint64_t lt;
    asm("s_mov_b64 exec, %4\\n\\tv_add_f64 %0, %3, |%5|\\n\\tv_cmp_lt_f64 %1, %0, %2\\n\\tv_min_f64 %2, %2, %0\\n\\ts_mov_b64 exec, -1"
        : "=&v"(t), "=&s"(lt), "+v"(best) : "v"(src), "s"(mask), "v"(a));
    return lt;
}}
__device__ __forceinline__ uint64_t grp_cm(double &best, double c, uint64_t mask)
{{
    uint64_t lt;
    asm("s_mov_b64 exec, %3\\n\\tv_cmp_lt_f64 %0, %2, %1\\n\\tv_min_f64 %1, %1, %2\\n\\ts_mov_b64 exec, -1"
        : "=&s"(lt), "+v"(best) : "v"(c), "s"(mask));
    return lt;
}}
template <int OFF> __device__ __forceinline__ void store1(uint64_t m, uint64_t *p) {{ asm volatile("s_store_dwordx2 %0, %1, %2" ::"s"(m), "s"(p), "n"(OFF) : "memory"); }}
template <int OFF> __device__ __forceinline__ void store2(uint64_t m0, uint64_t m1, uint64_t *p)
{{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = {{(uint32_t)m0, (uint32_t)(m0 >> 32), (uint32_t)m1, (uint32_t)(m1 >> 32)}};
    asm volatile("s_store_dwordx4 %0, %1, %2" ::"s"(v), "s"(p), "n"(OFF) : "memory");
}}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void fill_t(const double *sig, int n_reads, int T, uint64_t *bp, double *last_row, int stride, const unsigned char *mask)
{{
    const int lane = threadIdx.x, q = lane % G, g = lane / G;
    const int wave = blockIdx.x;
    int read = wave * RPW + g;
    const bool live = read < n_reads;
    if (!live) read = n_reads - 1;
    const double *sp = sig + (size_t)read * T;
    uint64_t *wbp = bp + (size_t)wave * T * NWP;
    {regs}
'''
    init = []
    for k in range(n):
        init.append(f'    v_{k} = vtab[q][{k}]; {{ const int j = jtab[q][{k}]; double d0 = kInf; if (j == 0) d0 = start; else if (j > 0 && j <= M) d0 = start + fabs(sp[j] - v0); D_{k} = d0; }} c1_{k} = kInf; c2_{k} = kInf; c3_{k} = kInf;')
    adv = []
    for k in range(n):
        adv.append(f'        {{ const double a = s - v_{k}; c3_{k} = add_abs(c2_{k}, a); if (MASKED) c3_{k} = add_abs_masked(c3_{k}, c1_{k}, a, mrow(i + 1)); c2_{k} = add_abs(c1_{k}, a); c1_{k} = add_abs(D_{k}, a); D_{k} = kInf; }}')
    outp = []
    for k in range(n):
        outp.append(f'        {{ const int j = jtab[q][{k}]; if (j >= 0) last_row[(size_t)read * stride + j] = D_{k}; }}')
    src += f'''    const double v0 = vtab[0][0];
    const double start = fabs(sp[0] - v0);
{chr(10).join(init)}
    const unsigned char *mp = mask ? mask + (size_t)read * T : nullptr;
    auto mrow = [&](int r) -> uint64_t {{ return MASKED ? __ballot(r < T && mp[r] != 0) : 0ull; }}; // the reads masked at row r
    for (int i = 1; i < M; i++) {{
        const double s = sp[i];
{chr(10).join(adv)}
    }}
    typedef double d2 __attribute__((ext_vector_type(2)));
    auto row = [&](double s, uint64_t *rowp, uint64_t mnext) __attribute__((always_inline)) {{
{body}
    }};
    int i = M;
    for (; i < T && (i & 7); i++) row(sp[i], wbp + (size_t)i * NWP, mrow(i + 1));
    // eight rows per round; the samples arrive two at a time, two loads in flight (8 VGPRs instead of 32 for whole rounds)
    auto pair = [&](int r) -> d2 {{ const int c = r + 2 <= T ? r : T - 2; return *(const d2 *)(sp + c); }};
    d2 p0 = pair(i), p1 = pair(i + 2);
    for (; i + 8 <= T; i += 8) {{
        uint64_t mm[8];
        if (MASKED) {{
            const int c = i + 9 <= T ? i + 1 : T - 8; // rows i+1 .. i+8 (clamped at the end of the read)
            const uint64_t bytes = *(const uint64_t *)(mp + c);
            const int sh = i + 1 - c;
#pragma unroll
            for (int u = 0; u < 8; u++) mm[u] = __ballot(u + sh < 8 && ((bytes >> (8 * ((u + sh) & 7))) & 0xffull) != 0);
        }} else {{
#pragma unroll
            for (int u = 0; u < 8; u++) mm[u] = 0;
        }}
        uint64_t *rp = wbp + (size_t)i * NWP;
        row(p0.x, rp, mm[0]); row(p0.y, rp + NWP, mm[1]); p0 = pair(i + 4);
        row(p1.x, rp + 2 * NWP, mm[2]); row(p1.y, rp + 3 * NWP, mm[3]); p1 = pair(i + 6);
        row(p0.x, rp + 4 * NWP, mm[4]); row(p0.y, rp + 5 * NWP, mm[5]); p0 = pair(i + 8);
        row(p1.x, rp + 6 * NWP, mm[6]); row(p1.y, rp + 7 * NWP, mm[7]); p1 = pair(i + 10);
    }}
    for (; i < T; i++) row(sp[i], wbp + (size_t)i * NWP, mrow(i + 1));
    if (live && last_row) {{
{chr(10).join(outp)}
    }}
    asm volatile("s_dcache_wb" ::: "memory");
}}
}} // namespace

extern "C" int fill_t_words() {{ return NWP; }}
extern "C" int fill_t_rpw() {{ return RPW; }}
extern "C" float run_fill_t(const double *sig, int n_reads, int T, unsigned long long *bp, double *last_row, int stride, int reps, const unsigned char *mask)
{{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int waves = (n_reads + RPW - 1) / RPW;
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {{
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(fill_t, dim3(waves), dim3(64), 0, 0, sig, n_reads, T, (uint64_t *)bp, last_row, stride, mask);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }}
    if (hipGetLastError() != hipSuccess) return -1.f;
    return best;
}}
'''
    with open(out, 'w') as fh:
        fh.write(src)
    print(f'S={S} G={G} n={n}: {NW} mask words per wave-row ({NW * 8 / RPW:.1f} B per read-row), {n_groups} groups, '
          f'{len(transports)} transports, {len(hoisted)} hoisted, VALU per wave-row {n_valu} = {n_valu * G / 64:.2f} per read-row')
    for wd in words:
        if not (wd[3] == 0 and wd[2] == wd[0] - 1 and len(wd[4]) == G):
            print('   group k=%d f=%d from kp=%d d=%d lanes %s' % wd)
    return dict(order=L, pos=pos, NW=NW)


def automaton(name):
    from warpstr_amd import synth
    import bench
    if name == 'headline':
        pattern, fl = bench.HEADLINE
        locus = synth.make_locus(pattern, fl, 2024, max_states=64)
    elif name == 'cfg1':
        pattern, fl, _ = bench.CFG1
        locus = synth.make_locus(pattern, fl, 1)
    else:
        raise SystemExit('unknown automaton')
    t = locus.template
    pp, pi = np.asarray(t.pred_ptr), np.asarray(t.pred_idx)
    preds = [list(map(int, pi[pp[j]:pp[j + 1]])) for j in range(t.n_states)]
    return locus, np.asarray(t.value), preds, int(t.endstate)


if __name__ == '__main__':
    name, G, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    _, values, preds, end = automaton(name)
    generate(values, preds, end, G, out, masked='--masked' in sys.argv)
