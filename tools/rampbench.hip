// tools/rampbench.hip -- what a launch costs AROUND its waves: kernels whose every wave runs a fixed instruction count (so a wave's
// life is known), launched with the grids the under-filled rollout uses, as 256- and 1024-thread workgroups, with and without LDS.
// Kernel time (events riding on the launch) minus one wave's life (its own s_memtime) = dispatch ramp + drain.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ unsigned long long g_first[2], g_mid[2], g_last[2];   // {start stamp (wall clock), life in shader cycles}
template <int BS>
__global__ __launch_bounds__(BS) void spin(double* out, int iters, int lds_doubles) {
    extern __shared__ double sh[];
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    if (lds_doubles) { for (int i = threadIdx.x; i < lds_doubles; i += BS) sh[i] = i; __syncthreads(); }
    double d0 = threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3;
    for (int i = 0; i < iters; ++i) {
        asm volatile("v_fma_f64 %0, %0, %4, %5\nv_fma_f64 %1, %1, %4, %5\nv_fma_f64 %2, %2, %4, %5\nv_fma_f64 %3, %3, %4, %5\n"
                     "v_fma_f64 %0, %0, %4, %5\nv_fma_f64 %1, %1, %4, %5\nv_fma_f64 %2, %2, %4, %5\nv_fma_f64 %3, %3, %4, %5\n"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(1.0000001), "v"(0.9999999));
    }
    if (lds_doubles) d0 += sh[threadIdx.x % lds_doubles];
    out[(size_t)blockIdx.x * BS + threadIdx.x] = d0 + d1 + d2 + d3;
    if (threadIdx.x == 0) {
        unsigned long long* g = blockIdx.x == 0 ? g_first : (blockIdx.x == gridDim.x - 1 ? g_last : (blockIdx.x == gridDim.x / 2 ? g_mid : nullptr));
        if (g) { g[0] = w0; g[1] = clock64() - c0; }
    }
}
int main() {
    double* out; CHK(hipMalloc(&out, sizeof(double) * 4096 * 1024));
    hipStream_t st; CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    int wall_khz = 100000; (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    printf("threads  blocks(x BS)  iters  lds_B | kernel us (events on the launch) | first/mid/last block: start offset us, life cycles\n");
    for (int lds : {0, 2000}) for (int iters : {0, 300, 1200}) for (int waves : {489 * 4, 977 * 4, 1954 * 4, 3908 * 4}) for (int bs : {256, 512, 1024}) {
        const int blocks = (waves * 64 + bs - 1) / bs;
        auto go = [&](bool timed) {
            const size_t ldsb = (size_t)lds * (bs / 256);
            if (bs == 256) hipExtLaunchKernelGGL(spin<256>, dim3(blocks), dim3(256), ldsb, st, timed ? a : nullptr, timed ? b : nullptr, 0, out, iters, (int)(ldsb / 8));
            else if (bs == 512) hipExtLaunchKernelGGL(spin<512>, dim3(blocks), dim3(512), ldsb, st, timed ? a : nullptr, timed ? b : nullptr, 0, out, iters, (int)(ldsb / 8));
            else hipExtLaunchKernelGGL(spin<1024>, dim3(blocks), dim3(1024), ldsb, st, timed ? a : nullptr, timed ? b : nullptr, 0, out, iters, (int)(ldsb / 8));
        };
        for (int w = 0; w < 3; ++w) go(false);
        CHK(hipStreamSynchronize(st));
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) { go(true); CHK(hipStreamSynchronize(st)); float ms; CHK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
        unsigned long long f[2], m[2], l[2];
        CHK(hipMemcpyFromSymbol(f, HIP_SYMBOL(g_first), sizeof(f))); CHK(hipMemcpyFromSymbol(m, HIP_SYMBOL(g_mid), sizeof(m))); CHK(hipMemcpyFromSymbol(l, HIP_SYMBOL(g_last), sizeof(l)));
        const double tick_us = 1e3 / wall_khz;
        printf("%5d %7d %6d %6d | %8.2f | first +0.00 life %6llu | mid %+7.2f life %6llu | last %+7.2f life %6llu\n", bs, blocks, iters, lds * (bs / 256), best * 1e3,
               f[1], ((double)m[0] - (double)f[0]) * tick_us, m[1], ((double)l[0] - (double)f[0]) * tick_us, l[1]);
    }
    // back-to-back dependent launches on one stream: the boundary cost
    for (int n : {1, 2, 4}) {
        CHK(hipStreamSynchronize(st));
        CHK(hipEventRecord(a, st));
        for (int r = 0; r < 50; ++r) for (int j = 0; j < n; ++j) hipLaunchKernelGGL(spin<256>, dim3(489), dim3(256), 0, st, out, 300, 0);
        CHK(hipEventRecord(b, st)); CHK(hipEventSynchronize(b));
        float ms; CHK(hipEventElapsedTime(&ms, a, b));
        printf("stream of %d x 50 launches of 489 x 256 (iters 300): %.2f us per launch\n", n, ms * 1e3 / (50 * n));
    }
    return 0;
}
