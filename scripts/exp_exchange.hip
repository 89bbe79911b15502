// Micro-benchmark for the predecessor exchange of the four-slot DTW row (dtw_fill_fast<4,4,2,1,..>: K = 4 slots, slot 0 takes
// two candidates, the others one; M = 4): the row's own arithmetic (31 VALU instructions: 5 candidate adds, 5 compares into
// SGPR pairs, 5 mins, 4 differences, 12 pipeline adds) with the chain predecessor (state j-1) fetched
//   MODE 0  through LDS as the product's slot-major kernel does it: every slot writes its export (4 ds_write_b64), every slot
//           reads its predecessor's (5 ds_read_b64, double buffered by row parity, issued a row ahead);
//   MODE 1  by DPP: two v_mov_b32 wave_shr:1 per slot for lanes 1..63, lane 0 patched from lane 63 of the slot below with two
//           v_mov_b32 wave_ror:1 issued first (the shift leaves lane 0 as it was); LDS only for slot 0's two reads, fed by the
//           writes of slots 0 and K-1;
//   MODE 2  DPP without the lane-0 patch (NOT a correct exchange: the floor of what any DPP route can cost);
//   MODE 3  lane-major: slot k >= 1 takes the export of slot k-1 of its own lane from a register (the product's LM kernels).
//   MODE 4  what the product's slot-major placement would allow without a new placement: every slot still writes and reads LDS
//           (chain starts may sit in any lane and read from anywhere), the chain predecessor of the other lanes comes by
//           v_cndmask_b32_dpp wave_shr:1 (shift and select in one instruction, two per slot);
//   MODE 5  as MODE 4 with the writes of slots 1 and 2 left out (a placement that keeps every LDS-read state in slots 0 and K-1).
// 4 waves per workgroup and 2 workgroups per CU (the product's occupancy: two waves per SIMD), 4 000 rows per wave.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/exp_exchange.hip -o build/exp/exp_exchange
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <type_traits>

#define AS4 __attribute__((address_space(4)))
constexpr int K = 4;
constexpr int EXW = K * 64 + 32;

__device__ __forceinline__ double add_abs(double x, double a)
{
    double r;
    asm("v_add_f64 %0, %1, |%2|" : "=v"(r) : "v"(x), "v"(a));
    return r;
}
__device__ __forceinline__ double min_f64(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint64_t lt_mask(double c, double b) { return __builtin_amdgcn_fcmp(c, b, 4); }

template <bool PATCH>
__device__ __forceinline__ double dpp_prev(double cur, double below)
{
    const long long c = __double_as_longlong(cur), b = __double_as_longlong(below);
    int lo = (int)(c & 0xffffffffll), hi = (int)(c >> 32);   // (no patch: lane 0 keeps its own value -- no extra move for `old`)
    if (PATCH) {   // every lane <- lane-1 of `below`, lane 0 <- its lane 63
        lo = __builtin_amdgcn_mov_dpp((int)(b & 0xffffffffll), 0x13C, 0xf, 0xf, false);
        hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), 0x13C, 0xf, 0xf, false);
    }
    lo = __builtin_amdgcn_update_dpp(lo, (int)(c & 0xffffffffll), 0x138, 0xf, 0xf, false);   // lanes 1..63 <- lane-1 of `cur`
    hi = __builtin_amdgcn_update_dpp(hi, (int)(c >> 32), 0x138, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// lanes of `fed` take the LDS value, the others the value of the lane before them in `cur`; lane 0 keeps the LDS value
__device__ __forceinline__ double sel_prev(double cur, double from_lds, uint64_t fed)
{
    const long long c = __double_as_longlong(cur), l = __double_as_longlong(from_lds);
    int lo = (int)(l & 0xffffffffll), hi = (int)(l >> 32);
    // (s_nop 1: a DPP source written by the VALU instruction before needs two wait states, and the assembler does not count
    // them inside inline asm)
    asm("s_mov_b64 vcc, %4\n\ts_nop 1\n\t"
        "v_cndmask_b32_dpp %0, %2, %0, vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %1, %3, %1, vcc wave_shr:1 row_mask:0xf bank_mask:0xf"
        : "+v"(lo), "+v"(hi) : "v"((int)(c & 0xffffffffll)), "v"((int)(c >> 32)), "s"(fed) : "vcc");
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <int MODE>
__global__ __launch_bounds__(256) void rowloop(const double *sig, double *out, unsigned long long *mout, int T)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    double *ex = lds + wib * 2 * EXW;
    const AS4 double *cs = (const AS4 double *)(sig + (size_t)(blockIdx.x * 4 + wib) * (T + 8));
    double v[K], g1[K], g2[K], g3[K], acur[K], e[2][K], e2[2], x[2][K];
    int paddr[K], paddr2;
    for (int k = 0; k < K; k++) {
        v[k] = 0.01 * (k * 64 + lane);
        g1[k] = 1.0 + k;
        g2[k] = 2.0 + k;
        g3[k] = 3.0 + k;
        acur[k] = 0.5;
        e[0][k] = e[1][k] = 1e9;
        x[0][k] = x[1][k] = 1e9;
        const int q = k * 64 + lane;
        paddr[k] = q == 0 ? K * 64 : q - 1;           // the chain predecessor's export slot
        ex[q] = 1e9;
        ex[EXW + q] = 1e9;
    }
    e2[0] = e2[1] = 1e9;
    uint64_t fed[K];
    for (int k = 0; k < K; k++) fed[k] = (0x0000000100000001ull << (k * 5)) | 1ull;   // three chain starts per slot
    paddr2 = (lane * 7 + 3) % (K * 64);               // slot 0's second candidate: somewhere else
    if (lane < 32) ex[K * 64 + lane] = ex[EXW + K * 64 + lane] = 1e9;
    unsigned long long m = 0;
    __builtin_amdgcn_wave_barrier();
    auto row = [&](auto parc, int i) __attribute__((always_inline)) {
        constexpr int par = decltype(parc)::value;
        const double snext = cs[i + 1];
        double *wb = ex + par * EXW, *rb = ex + (1 - par) * EXW;
        // exports of row i+1, issued a row ahead
        if (MODE == 0 || MODE >= 4) {
#pragma unroll
            for (int k = 0; k < K; k++) e[par][k] = rb[paddr[k]];
        } else {
            e[par][0] = rb[paddr[0]];
        }
        e2[par] = rb[paddr2];
        double enew[K];
#pragma unroll
        for (int kk = 0; kk < K; kk++) {
            const int k = MODE == 3 ? K - 1 - kk : kk;
            double best = g1[k];
            double src;
            if (MODE == 0 || k == 0) src = e[1 - par][k];
            else if (MODE >= 4) src = sel_prev(x[1 - par][k], e[1 - par][k], fed[k]);   // x: the slot's own export of the row before
            else if (MODE == 3) src = x[par][k - 1];
            else src = e[1 - par][k];   // (filled by the DPP moves at the end of the row before)
            double cand = add_abs(src, acur[k]);
            m ^= lt_mask(cand, best);
            best = min_f64(best, cand);
            if (k == 0) {
                cand = add_abs(e2[1 - par], acur[k]);
                m ^= lt_mask(cand, best);
                best = min_f64(best, cand);
            }
            const double an = snext - v[k];
            g3[k] = add_abs(g2[k], an);
            g2[k] = add_abs(g1[k], an);
            g1[k] = add_abs(best, an);
            acur[k] = an;
            enew[k] = g3[k];
            if (MODE == 3 && k < K - 1) x[par][k] = enew[k];
            if (MODE >= 4) x[par][k] = enew[k];
            if (MODE == 0 || MODE == 4 || k == 0 || k == K - 1) wb[k * 64 + lane] = enew[k];
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int k = 1; k < K; k++) e[par][k] = dpp_prev<MODE == 1>(enew[k], enew[k - 1]);
        }
        __builtin_amdgcn_wave_barrier();
    };
    for (int i = 0; i < T; i += 2) {
        row(std::integral_constant<int, 0>{}, i);
        row(std::integral_constant<int, 1>{}, i + 1);
    }
    double s = 0;
    for (int k = 0; k < K; k++) s += g1[k] + g2[k] + g3[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) mout[blockIdx.x * 4 + wib] = m;
}

template <int MODE>
void run(const double *sig, double *out, unsigned long long *mout, const char *name, int T)
{
    const int blocks = 256 * 2;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    const size_t shm = 4 * 2 * EXW * sizeof(double);
    hipLaunchKernelGGL(rowloop<MODE>, dim3(blocks), dim3(256), shm, 0, sig, out, mout, T);
    (void)hipEventRecord(a, 0);
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(rowloop<MODE>, dim3(blocks), dim3(256), shm, 0, sig, out, mout, T);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    ms /= 3;
    // 2 waves per SIMD, T rows each: ns per wave-row and SIMD
    printf("MODE %d  %-64s %8.3f ms   %.2f ns per wave-row and SIMD\n", MODE, name, ms, ms * 1e6 / (2.0 * T));
}

int main()
{
    const int T = 4000, blocks = 512;
    double *sig, *out;
    unsigned long long *mout;
    (void)hipMalloc(&sig, (size_t)blocks * 4 * (T + 8) * 8);
    (void)hipMemset(sig, 0, (size_t)blocks * 4 * (T + 8) * 8);
    (void)hipMalloc(&out, blocks * 256 * 8);
    (void)hipMalloc(&mout, blocks * 4 * 8);
    run<0>(sig, out, mout, "LDS, slot-major (4 ds_write_b64 + 5 ds_read_b64 per row)", T);
    run<1>(sig, out, mout, "DPP wave_shr:1 + lane-0 patch by wave_ror:1 (12 v_mov_b32_dpp per row)", T);
    run<2>(sig, out, mout, "DPP wave_shr:1 only, lane 0 wrong (6 v_mov_b32_dpp per row: the floor)", T);
    run<3>(sig, out, mout, "lane-major: registers of the own lane (2 ds_write_b64 + 2 ds_read_b64)", T);
    run<4>(sig, out, mout, "LDS as MODE 0 + v_cndmask_b32_dpp select (6 per row)", T);
    run<5>(sig, out, mout, "as MODE 4, slots 1 and 2 do not write (2 ds_write_b64 + 5 ds_read_b64)", T);
    return 0;
}
