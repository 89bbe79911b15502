// Experiment: do scalar stores (s_store_dwordx2/x4) work on gfx950, and is a v_cmp into an SGPR pair followed by a scalar
// store of that pair correct when the pair is reused every iteration?   hipcc --offload-arch=gfx950 -O3 exp_sstore.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__global__ void k(const double *x, uint64_t *out, int rows)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    uint64_t *o = out + (size_t)wave * rows * 2;
    double acc = x[lane] + wave;
    for (int i = 0; i < rows; i++) {
        const double t = x[(lane + i) & 63] * 3.0;
        const uint64_t m0 = __builtin_amdgcn_fcmp(t, acc, 4);       // t < acc, one bit per lane
        const uint64_t m1 = __builtin_amdgcn_fcmp(acc, t + 1.0, 4);
        const uint32_t off = (uint32_t)i * 16u;
        asm volatile("s_store_dwordx2 %0, %1, %2" ::"s"(m0), "s"(o), "s"(off) : "memory");
        asm volatile("s_store_dwordx2 %0, %1, %2" ::"s"(m1), "s"(o), "s"(off + 8u) : "memory");
        acc = acc * 0.5 + t;
    }
    asm volatile("s_dcache_wb" ::: "memory");
}

int main()
{
    const int rows = 2000, waves = 4096;
    std::vector<double> hx(64);
    for (int i = 0; i < 64; i++) hx[i] = (i * 37 % 64) / 7.0 - 3.0;
    double *dx;
    uint64_t *dout;
    hipMalloc(&dx, 64 * 8);
    hipMalloc(&dout, (size_t)waves * rows * 16);
    hipMemset(dout, 0xff, (size_t)waves * rows * 16);
    hipMemcpy(dx, hx.data(), 64 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(waves / 4), dim3(256), 0, 0, dx, dout, rows);
    hipError_t e = hipDeviceSynchronize();
    printf("sync: %s\n", hipGetErrorString(e));
    std::vector<uint64_t> h((size_t)waves * rows * 2);
    hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (int w = 0; w < waves; w++) {
        double acc[64];
        for (int l = 0; l < 64; l++) acc[l] = hx[l] + w;
        for (int i = 0; i < rows; i++) {
            uint64_t m0 = 0, m1 = 0;
            for (int l = 0; l < 64; l++) {
                const double t = hx[(l + i) & 63] * 3.0;
                if (t < acc[l]) m0 |= 1ull << l;
                if (acc[l] < t + 1.0) m1 |= 1ull << l;
                acc[l] = acc[l] * 0.5 + t;
            }
            if (h[((size_t)w * rows + i) * 2] != m0 || h[((size_t)w * rows + i) * 2 + 1] != m1) {
                if (bad < 5) printf("mismatch wave %d row %d: %016llx/%016llx vs %016llx/%016llx\n", w, i,
                                    (unsigned long long)h[((size_t)w * rows + i) * 2],
                                    (unsigned long long)h[((size_t)w * rows + i) * 2 + 1], (unsigned long long)m0,
                                    (unsigned long long)m1);
                bad++;
            }
        }
    }
    printf("scalar-store check: %zu mismatches of %zu records\n", bad, (size_t)waves * rows);
    return bad ? 1 : 0;
}
