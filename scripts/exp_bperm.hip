// Micro-benchmark: LDS-pipe cost of a 64-bit neighbour exchange on gfx950, three ways, with the VALU kept as busy as in the
// DTW fill (10 dependent-free fp64 ops per row):  (a) ds_write_b64 + 2 x ds_read_b64 (what dtw_fill_fast does),
// (b) 4 x ds_bpermute_b32 (no LDS memory), (c) no exchange at all (VALU floor).
//   hipcc --offload-arch=gfx950 -O3 exp_bperm.hip -o exp_bperm && ./exp_bperm
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(const double *x, double *out, const int *src0, const int *src1, int rows)
{
    __shared__ double lds[4][2][96];
    __shared__ __attribute__((aligned(16))) double lds2[4][192];
    double keep0 = 0.5, keep1 = 0.25;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double d = x[lane], g1 = d + 1.0, g2 = d + 2.0, g3 = d + 3.0, e0 = 0.5, e1 = 0.25;
    const double v = x[(lane * 7) & 63];
    const int a0 = src0[lane], a1 = src1[lane];
    for (int i = 0; i < rows; i++) {
        const int par = i & 1;
        double n0 = e0, n1 = e1;
        if (MODE == 0) {
            n0 = lds[w][1 - par][a0];
            n1 = lds[w][1 - par][a1];
        } else if (MODE == 3) { // one read only
            n0 = lds[w][1 - par][a0];
        } else if (MODE == 4) { // second read for the four lanes that have a second predecessor
            n0 = lds[w][1 - par][a0];
            if (a1 != 63) n1 = lds[w][1 - par][a1];
        } else if (MODE == 6) { // pairs of rows: 16-byte reads every other row
            typedef double d2 __attribute__((ext_vector_type(2)));
            if (par) {
                const d2 p0 = *(const d2 *)&lds2[w][a0 * 2];
                const d2 p1 = *(const d2 *)&lds2[w][a1 * 2];
                n0 = p0.x; keep0 = p0.y;
                n1 = p1.x; keep1 = p1.y;
            } else {
                n0 = keep0;
                n1 = keep1;
            }
        } else if (MODE == 5) { // write only
        } else if (MODE == 1) {
            const long long b = __double_as_longlong(g3);
            const int lo = (int)b, hi = (int)(b >> 32);
            const int l0 = __builtin_amdgcn_ds_bpermute(a0 * 4, lo), h0 = __builtin_amdgcn_ds_bpermute(a0 * 4, hi);
            const int l1 = __builtin_amdgcn_ds_bpermute(a1 * 4, lo), h1 = __builtin_amdgcn_ds_bpermute(a1 * 4, hi);
            n0 = __longlong_as_double(((long long)h0 << 32) | (unsigned)l0);
            n1 = __longlong_as_double(((long long)h1 << 32) | (unsigned)l1);
        }
        const double a = (double)i * 1e-9 - v;
        const double c0 = e0 + fabs(a), c1 = e1 + fabs(a);
        double best = g1;
        best = fmin(best, c0);
        best = fmin(best, c1);
        const double an = a + 1e-7;
        g3 = g2 + fabs(an);
        g2 = g1 + fabs(an);
        g1 = best + fabs(an);
        d = best;
        if (MODE == 0 || (MODE >= 3 && MODE != 6)) lds[w][par][lane] = g3;
        if (MODE == 6 && !par) {
            typedef double d2 __attribute__((ext_vector_type(2)));
            d2 pr = {g3, g2 + 1e-3};
            *(d2 *)&lds2[w][lane * 2] = pr;
        }
        e0 = n0;
        e1 = n1;
        __builtin_amdgcn_wave_barrier();
    }
    out[blockIdx.x * 256 + threadIdx.x] = d + g2;
}

template <int MODE>
float run(const double *dx, double *dout, const int *s0, const int *s1, int waves, int rows)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(waves / 4), dim3(256), 0, 0, dx, dout, s0, s1, rows);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(waves / 4), dim3(256), 0, 0, dx, dout, s0, s1, rows);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main()
{
    const int waves = 100000, rows = 2000;
    double hx[64];
    int h0[64], h1[64];
    for (int i = 0; i < 64; i++) {
        hx[i] = (i * 37 % 64) / 7.0 + 1.0;
        h0[i] = i ? i - 1 : 63;
        h1[i] = (i == 21 || i == 26 || i == 40 || i == 45) ? i - 3 : 63;
    }
    double *dx, *dout;
    int *s0, *s1;
    hipMalloc(&dx, 512);
    hipMalloc(&dout, (size_t)waves * 64 * 8);
    hipMalloc(&s0, 256);
    hipMalloc(&s1, 256);
    hipMemcpy(dx, hx, 512, hipMemcpyHostToDevice);
    hipMemcpy(s0, h0, 256, hipMemcpyHostToDevice);
    hipMemcpy(s1, h1, 256, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; rep++) {
        printf("lds write+2 reads : %.3f ms\n", run<0>(dx, dout, s0, s1, waves, rows));
        printf("4 x ds_bpermute   : %.3f ms\n", run<1>(dx, dout, s0, s1, waves, rows));
        printf("no exchange       : %.3f ms\n", run<2>(dx, dout, s0, s1, waves, rows));
        printf("write + 1 read    : %.3f ms\n", run<3>(dx, dout, s0, s1, waves, rows));
        printf("write + 1 read + 4-lane read : %.3f ms\n", run<4>(dx, dout, s0, s1, waves, rows));
        printf("write only        : %.3f ms\n", run<5>(dx, dout, s0, s1, waves, rows));
        printf("16-byte write + 2 reads every other row : %.3f ms\n", run<6>(dx, dout, s0, s1, waves, rows));
    }
    return 0;
}
