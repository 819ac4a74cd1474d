// which SIMD each wave of a 384-thread (six-wave) workgroup lands on when a workgroup owns its CU (150 KB of LDS)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512) void k(unsigned* out) {
    extern __shared__ char smem[];
    smem[threadIdx.x] = 1;
    const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all 32 bits
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw;
}
int main() {
    for (int waves : {4, 6, 8}) {
        unsigned* d; hipMalloc(&d, 512 * 8 * 4); hipMemset(d, 0xff, 512 * 8 * 4);
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 150 * 1024, 0, d);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        std::vector<unsigned> h(512 * 8); hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        int hist[8][4] = {};
        for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) hist[w][(h[b * 8 + w] >> 4) & 3]++;
        printf("waves per workgroup %d: wave -> SIMD histogram over 256 workgroups\n", waves);
        for (int w = 0; w < waves; ++w) printf("  wave %d: simd0 %d simd1 %d simd2 %d simd3 %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
        for (int b = 0; b < 3; ++b) { printf("  block %d:", b); for (int w = 0; w < waves; ++w) printf(" %08x", h[b * 8 + w]); printf("\n"); }
        hipFree(d);
    }
    return 0;
}
