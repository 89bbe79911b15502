// Micro-benchmark: what makes a 64-bit LDS access of a wavefront conflict-free on gfx950?  One wave per SIMD slot writes
// ds_write_b64 / reads ds_read_b64 through per-lane slot tables (slot = 8-byte word index); time per access for several tables.
//   hipcc --offload-arch=gfx950 -O3 scripts/exp_ldsbank.hip -o /tmp/exp_ldsbank && /tmp/exp_ldsbank
#include <hip/hip_runtime.h>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

template <bool W, bool R>
__global__ __launch_bounds__(256) void k(const int *wtab, const int *rtab, double *out, int iters)
{
    __shared__ double lds[4][2][128];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int ws = wtab[lane], rs = rtab[lane];
    double v = lane * 0.5, acc = 0.0;
    for (int q = lane; q < 128; q += 64) lds[w][0][q] = lds[w][1][q] = 1.0;
    for (int i = 0; i < iters; i++) {
        const int par = i & 1;
        if (W) lds[w][par][ws] = v;
        if (R) acc += lds[w][1 - par][rs];
        v += 1.0;
        __builtin_amdgcn_wave_barrier();
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc + v;
}

template <bool W, bool R>
float run(const int *dw, const int *dr, double *dout, int iters)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int blocks = 256 * 8;
    hipLaunchKernelGGL((k<W, R>), dim3(blocks), dim3(256), 0, 0, dw, dr, dout, iters);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((k<W, R>), dim3(blocks), dim3(256), 0, 0, dw, dr, dout, iters);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main()
{
    const int iters = 20000;
    int *dw, *dr;
    double *dout;
    hipMalloc(&dw, 256);
    hipMalloc(&dr, 256);
    hipMalloc(&dout, 256 * 8 * 256 * 8);
    struct T { std::string name; std::function<int(int)> f; };
    std::vector<T> tabs = {
        {"identity            slot = lane", [](int l) { return l; }},
        {"chain               slot = lane-1 (mod 64)", [](int l) { return (l + 63) % 64; }},
        {"shift 4             slot = lane-4 (mod 64): crosses the halves", [](int l) { return (l + 60) % 64; }},
        {"swap halves         slot = lane ^ 32", [](int l) { return l ^ 32; }},
        {"permute in half     slot = half*32 + (5*lane+3) mod 32", [](int l) { return (l / 32) * 32 + (5 * l + 3) % 32; }},
        {"reverse in half     slot = half*32 + 31 - lane%32", [](int l) { return (l / 32) * 32 + 31 - l % 32; }},
        {"two halves, 1 bank pair set: slot = 2*(lane%32) + lane/32 (even/odd interleave)", [](int l) { return 2 * (l % 32) + l / 32; }},
        {"2-way conflict      slot = (lane%32)/2*2 + 32*(lane&1) + 64*(lane/32)", [](int l) { return (l % 32) / 2 * 2 + 32 * (l & 1) + 64 * (l / 32); }},
        {"stride 2 slots      slot = 2*lane (mod 128)", [](int l) { return (2 * l) % 128; }},
        {"quarter swap        slot = lane ^ 16", [](int l) { return l ^ 16; }},
        {"lanes 0-3 far, rest shifted: 0..3 -> 57,18,52,33; l -> l-4", [](int l) { static const int m[4] = {57, 18, 52, 33}; if (l < 4) return m[l]; int s = l - 4; while (s == 57 || s == 18 || s == 52 || s == 33) s = 60 + (s % 4); return s; }},
        {"broadcast           slot = 5", [](int) { return 5; }},
        {"A  8 lanes distinct mod 8, but 16 lanes collide mod 16: lanes 8..15 -> 16..23", [](int l) { return (l & 7) + 16 * ((l >> 3) & 1) + 32 * (l >> 4 & 1) + 8 * (l >> 5); }},
        {"B  16 lanes distinct mod 16, but 8 lanes collide mod 8: 0,8,1,9,..", [](int l) { const int q = l & 15; return (l & 48) + (q >> 1) + 8 * (q & 1); }},
        {"C  8 lanes distinct mod 8, 4 lanes collide mod 4: 0,4,1,5,2,6,3,7", [](int l) { const int q = l & 7; return (l & 56) + (q >> 1) + 4 * (q & 1); }},
        {"D  packed layout after the quarter rule (wsx_place.h, 16-lane rule only)", [](int l) { static const int t[64] = {57, 18, 52, 33, 0, 3, 5, 6, 7, 8, 10, 11, 12, 13, 14, 15, 2, 4, 9, 16, 49, 19, 21, 22, 23, 24, 26, 27, 28, 29, 30, 31, 1, 32, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 20, 25, 48, 17, 50, 51, 53, 54, 55, 56, 58, 59, 60, 61, 62, 63}; return t[l]; }},
        {"E  lanes 0..7 -> 8,17,26,35,44,53,62,7 (distinct mod 8 and mod 16), rest identity-like", [](int l) { static const int t[8] = {8, 17, 26, 35, 44, 53, 62, 7}; if (l < 8) return t[l]; int s = l; return (s == 8 || s == 17 || s == 26 || s == 35 || s == 44 || s == 53 || s == 62) ? 64 + (l & 7) + 8 * (l >> 3) : s; }},
        {"T1 identity, lanes 0 and 1 trade slots", [](int l) { return l == 0 ? 1 : l == 1 ? 0 : l; }},
        {"T2 identity, lanes 0/1 and 32/33 trade slots (lane l and l+32 keep one bank pair)", [](int l) { return (l & 31) == 0 ? l + 1 : (l & 31) == 1 ? l - 1 : l; }},
        {"T3 scrambled half, the same scramble in both halves: slot = half*32 + (11*l*l+7*l+5) perm", [](int l) { static const int p[32] = {5, 23, 9, 30, 1, 17, 12, 28, 3, 21, 14, 26, 7, 19, 0, 31, 10, 24, 2, 16, 13, 29, 6, 20, 11, 27, 4, 18, 15, 25, 8, 22}; return (l / 32) * 32 + p[l % 32]; }},
        {"T4 scrambled halves, different scrambles", [](int l) { static const int p[32] = {5, 23, 9, 30, 1, 17, 12, 28, 3, 21, 14, 26, 7, 19, 0, 31, 10, 24, 2, 16, 13, 29, 6, 20, 11, 27, 4, 18, 15, 25, 8, 22}; return l < 32 ? p[l] : 32 + p[(l * 7 + 3) % 32]; }},
        {"T5 identity in the low half, high half rotated by one", [](int l) { return l < 32 ? l : 32 + (l + 1) % 32; }},
        {"T6 T3 with the halves' slots swapped for odd lanes (lane l, l+32: one bank pair, either half)", [](int l) { static const int p[32] = {5, 23, 9, 30, 1, 17, 12, 28, 3, 21, 14, 26, 7, 19, 0, 31, 10, 24, 2, 16, 13, 29, 6, 20, 11, 27, 4, 18, 15, 25, 8, 22}; int h = l / 32; if (l & 1) h ^= 1; return h * 32 + p[l % 32]; }},
        {"T7 lanes 0..15 and 16..31 trade places blockwise, high half identity", [](int l) { return l < 32 ? (l ^ 16) : l; }},
        {"T8 identity in low half, high half: pairs of lanes trade slots", [](int l) { return l < 32 ? l : l ^ 1; }},
        {"headline packed layout: export slots written (wsx_place.h)", [](int l) { static const int t[64] = {57, 18, 52, 33, 0, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 49, 19, 21, 22, 23, 24, 26, 27, 28, 29, 30, 31, 1, 20, 25, 32, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 17, 50, 51, 53, 54, 55, 56, 58, 59, 60, 61, 62, 63}; return t[l]; }},
        {"headline packed layout: first-predecessor slots read", [](int l) { static const int t[64] = {56, 17, 51, 32, 95, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 48, 18, 20, 21, 22, 23, 25, 26, 27, 28, 29, 30, 0, 19, 24, 31, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 16, 49, 31, 52, 53, 53, 16, 57, 58, 58, 82, 82, 82}; return t[l]; }},
        {"headline packed layout: second-predecessor slots read", [](int l) { static const int t[4] = {59, 60, 54, 55}; return l < 4 ? t[l] : 64; }},
        {"natural layout: first-predecessor slots read (lane-1, lane 0 reads a spare slot)", [](int l) { return l ? l - 1 : 95; }},
        {"natural layout: second-predecessor slots read (4 lanes, rest one spare slot)", [](int l) { return l == 18 ? 16 : l == 23 ? 22 : l == 38 ? 36 : l == 43 ? 42 : 64; }},
    };
    std::vector<int> id(64);
    for (int l = 0; l < 64; l++) id[l] = l;
    printf("%-90s %10s %10s\n", "table", "write ns", "read ns");
    const double waves_per_simd = 8.0 * 4 / 4; // 8 blocks of 4 waves per CU = 8 waves per SIMD
    for (auto &t : tabs) {
        std::vector<int> h(64);
        for (int l = 0; l < 64; l++) h[l] = t.f(l);
        hipMemcpy(dw, h.data(), 256, hipMemcpyHostToDevice);
        hipMemcpy(dr, id.data(), 256, hipMemcpyHostToDevice);
        const float wms = run<true, false>(dw, dr, dout, iters);
        hipMemcpy(dw, id.data(), 256, hipMemcpyHostToDevice);
        hipMemcpy(dr, h.data(), 256, hipMemcpyHostToDevice);
        const float rms = run<false, true>(dw, dr, dout, iters);
        // per CU the LDS pipe serves 32 waves x iters accesses in the measured time
        printf("%-90s %10.2f %10.2f   (ns per wave access on one CU's pipe)\n", t.name.c_str(), wms * 1e6 / (iters * 32.0), rms * 1e6 / (iters * 32.0));
    }
    (void)waves_per_simd;
    return 0;
}
