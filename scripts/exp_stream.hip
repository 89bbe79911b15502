// Experiment: bandwidth of "one thread streams its own contiguous region" (the access pattern of a thread-per-read
// traceback over per-row records) against "one wave streams a region" (wave-per-read).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

// thread t reads region t (bytes_per bytes, descending addresses), LINES x 64 B per iteration
template <int LINES>
__global__ __launch_bounds__(64) void per_thread(const u4 *buf, uint32_t *out, int n, int bytes_per)
{
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= n) return;
    const u4 *p = buf + (size_t)t * (bytes_per / 16);
    uint32_t acc = 0;
    for (int o = bytes_per / 16 - 4 * LINES; o >= 0; o -= 4 * LINES) {
#pragma unroll
        for (int q = 0; q < 4 * LINES; q++) {
            const u4 v = p[o + q];
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    out[t] = acc;
}

// wave w reads region w, 1 KiB per wave-instruction
__global__ __launch_bounds__(256) void per_wave(const u4 *buf, uint32_t *out, int n, int bytes_per)
{
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= n) return;
    const u4 *p = buf + (size_t)w * (bytes_per / 16);
    uint32_t acc = 0;
    for (int o = bytes_per / 16 - 128; o >= 0; o -= 128) {
        const u4 a = p[o + lane], b = p[o + 64 + lane];
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
    }
    if (acc == 0x12345678u) out[w] = acc;
}

int main()
{
    const int n = 100000, bytes_per = 32768;
    u4 *buf;
    uint32_t *out;
    hipMalloc(&buf, (size_t)n * bytes_per);
    hipMalloc(&out, n * 4);
    hipMemset(buf, 1, (size_t)n * bytes_per);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto run = [&](const char *name, auto launch) {
        float best = 1e9f;
        for (int r = 0; r < 4; r++) {
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%-28s %.3f ms  %.2f TB/s\n", name, best, (double)n * bytes_per / best / 1e9);
    };
    run("thread per region, 64 B/iter", [&] { hipLaunchKernelGGL(per_thread<1>, dim3((n + 63) / 64), dim3(64), 0, 0, buf, out, n, bytes_per); });
    run("thread per region, 128 B/iter", [&] { hipLaunchKernelGGL(per_thread<2>, dim3((n + 63) / 64), dim3(64), 0, 0, buf, out, n, bytes_per); });
    run("thread per region, 256 B/iter", [&] { hipLaunchKernelGGL(per_thread<4>, dim3((n + 63) / 64), dim3(64), 0, 0, buf, out, n, bytes_per); });
    run("wave per region, 2 KiB/iter", [&] { hipLaunchKernelGGL(per_wave, dim3((n + 3) / 4), dim3(256), 0, 0, buf, out, n, bytes_per); });
    printf("%s\n", hipGetErrorString(hipDeviceSynchronize()));
    return 0;
}
