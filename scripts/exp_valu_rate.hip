// Micro-benchmark: issue cost of the fp64 VALU instructions the DTW fill consists of, on gfx950.  Every SIMD gets 8 waves
// that run long blocks of independent instructions of one kind; cycles per wave-instruction and SIMD come from
// GRBM_GUI_ACTIVE and SQ_INSTS_VALU (run under rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU: scripts/exp_valu_rate.sh).
//   hipcc --offload-arch=gfx950 -O3 scripts/exp_valu_rate.hip -o build/exp/exp_valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X X X X X X X X
template <int KIND>
__global__ __launch_bounds__(256) void k(double *out, int iters, double seed)
{
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double b = seed * 0.5;
    unsigned long long m = 0;
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) { // v_add_f64 with |src| modifier (the fill's add)
            REP8(asm volatile("v_add_f64 %0, %0, |%8|\n v_add_f64 %1, %1, |%8|\n v_add_f64 %2, %2, |%8|\n v_add_f64 %3, %3, |%8|\n"
                              "v_add_f64 %4, %4, |%8|\n v_add_f64 %5, %5, |%8|\n v_add_f64 %6, %6, |%8|\n v_add_f64 %7, %7, |%8|"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if (KIND == 1) { // v_min_f64
            REP8(asm volatile("v_min_f64 %0, %0, %8\n v_min_f64 %1, %1, %8\n v_min_f64 %2, %2, %8\n v_min_f64 %3, %3, %8\n"
                              "v_min_f64 %4, %4, %8\n v_min_f64 %5, %5, %8\n v_min_f64 %6, %6, %8\n v_min_f64 %7, %7, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if (KIND == 2) { // v_cmp_lt_f64 into an SGPR pair (VOP3), the fill's back-pointer compare
            unsigned long long s0, s1, s2, s3;
            REP8(asm volatile("v_cmp_lt_f64 %0, %4, %8\n v_cmp_lt_f64 %1, %5, %8\n v_cmp_lt_f64 %2, %6, %8\n v_cmp_lt_f64 %3, %7, %8\n"
                              "v_cmp_lt_f64 %0, %6, %8\n v_cmp_lt_f64 %1, %7, %8\n v_cmp_lt_f64 %2, %4, %8\n v_cmp_lt_f64 %3, %5, %8"
                              : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b));
                 m += s0 ^ s1 ^ s2 ^ s3;)
        } else if (KIND == 3) { // v_cmp_lt_f64 into vcc (VOPC)
            REP8(asm volatile("v_cmp_lt_f64 vcc, %0, %4\n v_cmp_lt_f64 vcc, %1, %4\n v_cmp_lt_f64 vcc, %2, %4\n v_cmp_lt_f64 vcc, %3, %4\n"
                              "v_cmp_lt_f64 vcc, %0, %4\n v_cmp_lt_f64 vcc, %1, %4\n v_cmp_lt_f64 vcc, %2, %4\n v_cmp_lt_f64 vcc, %3, %4"
                              :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b) : "vcc");)
        } else if (KIND == 4) { // v_add_f64 with an SGPR operand (signal sample): s - v
            double sc = seed;
            asm volatile("" : "+s"(sc));
            REP8(asm volatile("v_add_f64 %0, %8, -%0\n v_add_f64 %1, %8, -%1\n v_add_f64 %2, %8, -%2\n v_add_f64 %3, %8, -%3\n"
                              "v_add_f64 %4, %8, -%4\n v_add_f64 %5, %8, -%5\n v_add_f64 %6, %8, -%6\n v_add_f64 %7, %8, -%7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sc));)
        } else { // the fill's row mix, dependent as in the kernel: 6 adds, 2 compares, 2 mins
            unsigned long long s0, s1;
            REP8(asm volatile("v_add_f64 %2, %4, |%8|\n v_cmp_lt_f64 %0, %2, %5\n v_min_f64 %2, %5, %2\n v_add_f64 %3, %6, |%8|\n"
                              "v_cmp_lt_f64 %1, %3, %2\n v_min_f64 %3, %2, %3\n v_add_f64 %4, %9, -%8\n v_add_f64 %5, %5, |%4|\n"
                              "v_add_f64 %6, %6, |%4|\n v_add_f64 %7, %3, |%4|"
                              : "=s"(s0), "=s"(s1), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(a6), "v"(b));
                 m += s0 ^ s1;)
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (double)m;
}

template <int KIND>
void run(double *dout, const char *name, int per_iter)
{
    const int iters = 4000;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<KIND>, dim3(256 * 8), dim3(256), 0, 0, dout, iters, 1.5);
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL(k<KIND>, dim3(256 * 8), dim3(256), 0, 0, dout, iters, 1.5);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    // 8 waves per SIMD, each issues per_iter * iters instructions of this kind
    printf("%-56s %8.3f ms  %.3f ns per wave-instruction and SIMD\n", name, ms, ms * 1e6 / (8.0 * per_iter * iters));
}

int main()
{
    double *dout;
    (void)hipMalloc(&dout, 256 * 8 * 256 * 8);
    run<0>(dout, "v_add_f64 d, d, |v|", 64);
    run<1>(dout, "v_min_f64", 64);
    run<2>(dout, "v_cmp_lt_f64 sgpr-pair, v, v (VOP3)", 64);
    run<3>(dout, "v_cmp_lt_f64 vcc, v, v (VOPC)", 64);
    run<4>(dout, "v_add_f64 d, sgpr, -v", 64);
    run<5>(dout, "fill row mix (6 add, 2 cmp->sgpr, 2 min), dependent", 80);
    return 0;
}
