// Micro-benchmark: what does a ds_write_b64 cost when only some lanes of the wavefront are active?  (Question behind the
// lane-major fills that export every slot: could slots export only from the lanes somebody reads?)  Four writes per
// iteration, each under the same 64-bit lane mask; 4 waves per block, every SIMD slot busy; time per wave-iteration.
//   hipcc --offload-arch=gfx950 -O3 scripts/exp_ldsmask.hip -o /tmp/exp_ldsmask && /tmp/exp_ldsmask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void k(uint64_t mask, double *out, int iters, int nwrites)
{
    __shared__ double lds[4][2][4 * 64 + 32];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool on = (mask >> lane) & 1ull;
    double v0 = lane * 0.5, v1 = lane * 0.25, v2 = lane * 0.125, v3 = lane * 2.0, acc = 0.0;
    for (int q = lane; q < 4 * 64 + 32; q += 64) lds[w][0][q] = lds[w][1][q] = 1.0;
    for (int i = 0; i < iters; i++) {
        const int par = i & 1;
        if (on) {
            lds[w][par][lane] = v0;
            if (nwrites > 1) lds[w][par][64 + lane] = v1;
            if (nwrites > 2) lds[w][par][128 + lane] = v2;
            if (nwrites > 3) lds[w][par][192 + lane] = v3;
        }
        acc += lds[w][1 - par][(lane + 1) & 63]; // one read per iteration, as slot 0 of the fill does
        // ~31 fp64 VALU instructions of filler would hide the LDS pipe; here the LDS pipe is what is measured: little filler
        v0 += 1.0;
        v1 += acc;
        v2 += 1.0;
        v3 += 1.0;
        __builtin_amdgcn_wave_barrier();
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc + v0 + v1 + v2 + v3;
}

static float run(uint64_t mask, double *dout, int iters, int nwrites)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int blocks = 256 * 8;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, mask, dout, iters, nwrites);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, mask, dout, iters, nwrites);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main()
{
    const int iters = 20000;
    double *dout;
    hipMalloc(&dout, 256 * 8 * 256 * 8);
    struct { const char *name; uint64_t m; } masks[] = {
        {"no lane", 0ull}, {"lane 0", 1ull}, {"lanes 0-7", 0xffull}, {"lanes 0-15", 0xffffull}, {"lanes 0-31 (low half)", 0xffffffffull},
        {"every second lane", 0x5555555555555555ull}, {"every fourth lane", 0x1111111111111111ull},
        {"lanes 0-7 and 32-39", 0x000000ff000000ffull}, {"all 64 lanes", ~0ull}};
    // waves per CU: 4 SIMDs x 8 blocks x ... the kernel is launched with 2048 blocks of 4 waves; per-CU LDS time per wave-iteration
    // = total time / (iterations x waves per CU in sequence): reported as ns per wave-iteration of one CU's LDS pipe share
    for (int nw = 1; nw <= 4; nw += 3)
        for (auto &m : masks) {
            const float ms = run(m.m, dout, iters, nw);
            const double waves_per_cu = 2048.0 * 4 / 256;
            printf("%d write(s) per iteration, %-24s %8.3f ms  = %6.2f ns per wave-iteration on its CU\n", nw, m.name, ms,
                   ms * 1e6 / (iters * waves_per_cu));
        }
    return 0;
}
