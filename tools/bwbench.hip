// tools/bwbench.hip -- HBM read / write / copy ceilings on this box for the access shapes the MPPI kernels use.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(256) void read4(const float4* __restrict__ p, size_t n4, float* out) {
    float acc = 0.f;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    for (; i + stride < n4; i += 2 * stride) { float4 a = p[i], b = p[i + stride]; acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w; }
    for (; i < n4; i += stride) { float4 a = p[i]; acc += a.x + a.y + a.z + a.w; }
    if (acc == 12345.678f) out[0] = acc;
}
// contiguous chunk per block (like update_kernel): block b reads [b*chunk, (b+1)*chunk)
__global__ __launch_bounds__(256) void read4_chunk(const float4* __restrict__ p, size_t n4, size_t chunk4, float* out) {
    float acc = 0.f;
    size_t lo = (size_t)blockIdx.x * chunk4, hi = lo + chunk4 < n4 ? lo + chunk4 : n4;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) { float4 a = p[i]; acc += a.x + a.y + a.z + a.w; }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void write1(float* __restrict__ p, size_t n, int rows) {  // 4 B/lane stores, row-strided like rollout
    size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; size_t K = n / rows;
    if (k >= K) return;
    for (int r = 0; r < rows; ++r) p[(size_t)r * K + k] = (float)r;
}
__global__ __launch_bounds__(256) void write4(float4* __restrict__ p, size_t n4) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    for (; i < n4; i += stride) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
int main() {
    const size_t bytes = 600ull << 20; const size_t n = bytes / 4, n4 = n / 4;
    float* buf; float* out; CHK(hipMalloc(&buf, bytes)); CHK(hipMalloc(&out, 4)); CHK(hipMemset(buf, 0, bytes));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    auto run = [&](const char* name, auto launch) { for (int w = 0; w < 2; ++w) launch(); hipDeviceSynchronize(); hipEventRecord(a, 0); for (int i = 0; i < 10; ++i) launch(); hipEventRecord(b, 0); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); printf("%-40s %8.1f us  %7.1f GB/s\n", name, ms * 100, bytes / (ms * 1e-4) / 1e9); };
    for (int blocks : {2048, 4096, 8192, 16384}) { char nm[64]; snprintf(nm, 64, "read float4 grid-stride x2, %d blocks", blocks); run(nm, [&] { hipLaunchKernelGGL(read4, dim3(blocks), dim3(256), 0, 0, (const float4*)buf, n4, out); }); }
    for (int blocks : {2048, 4096, 8192, 16384}) { char nm[64]; snprintf(nm, 64, "read float4 chunked, %d blocks", blocks); size_t c = (n4 + blocks - 1) / blocks; run(nm, [&] { hipLaunchKernelGGL(read4_chunk, dim3(blocks), dim3(256), 0, 0, (const float4*)buf, n4, c, out); }); }
    run("write 4B/lane, 150 rows x 1M (rollout)", [&] { hipLaunchKernelGGL(write1, dim3((1048576 + 255) / 256), dim3(256), 0, 0, buf, (size_t)150 * 1048576, 150); });
    for (int blocks : {2048, 8192}) { char nm[64]; snprintf(nm, 64, "write float4 grid-stride, %d blocks", blocks); run(nm, [&] { hipLaunchKernelGGL(write4, dim3(blocks), dim3(256), 0, 0, (float4*)buf, n4); }); }
    // Infinity Cache (256 MB) residency: the same chunked read over smaller buffers, and write-then-read
    for (size_t mb : {32, 64, 128, 200, 256, 400}) {
        const size_t b2 = mb << 20, m4 = b2 / 16; const int blocks = 8192; size_t c = (m4 + blocks - 1) / blocks;
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(read4_chunk, dim3(blocks), dim3(256), 0, 0, (const float4*)buf, m4, c, out);
        hipDeviceSynchronize(); hipEventRecord(a, 0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(read4_chunk, dim3(blocks), dim3(256), 0, 0, (const float4*)buf, m4, c, out);
        hipEventRecord(b, 0); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
        printf("re-read %4zu MB (chunked float4)            %8.1f us  %7.1f GB/s\n", mb, ms * 100, b2 / (ms * 1e-4) / 1e9);
        // write (4 B/lane rows) then read back once: what the update kernel sees after the rollout
        float tw = 0, tr = 0;
        for (int i = 0; i < 5; ++i) {
            hipEventRecord(a, 0);
            hipLaunchKernelGGL(write1, dim3((1048576 + 255) / 256), dim3(256), 0, 0, buf, b2 / 4, (int)(b2 / 4 / 1048576));
            hipEventRecord(b, 0); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b); tw += ms;
            hipEventRecord(a, 0);
            hipLaunchKernelGGL(read4_chunk, dim3(blocks), dim3(256), 0, 0, (const float4*)buf, m4, c, out);
            hipEventRecord(b, 0); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b); tr += ms;
        }
        printf("write %4zu MB then read it                  write %7.1f GB/s  read %7.1f GB/s\n", mb, b2 / (tw / 5 * 1e-3) / 1e9, b2 / (tr / 5 * 1e-3) / 1e9);
    }
    return 0;
}
