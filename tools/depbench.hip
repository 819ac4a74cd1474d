// tools/depbench.hip -- does a DEPENDENT VALU instruction issue behind its producer without a stall on gfx950?
// One wave's own view (s_memtime around its loop) of chains of 64 instructions per iteration: 8 independent chains against ONE
// dependent chain, at 1, 2 and 4 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 tools/depbench.hip -o tools/depbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ unsigned long long g_clk[2];
#define BEGIN unsigned long long c0__ = clock64(), w0__ = wall_clock64();
#define END if (blockIdx.x == 0 && threadIdx.x == 0) { g_clk[0] = clock64() - c0__; g_clk[1] = wall_clock64() - w0__; }
#define R8(x) x x x x x x x x
#define IND64(NAME, T, INIT, OP8)                                                                                      \
__global__ void NAME(double* out, int iters) {                                                                          \
    T d0 = INIT, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;             \
    const T k = (T)1.0000001, m = (T)0.9999999;                                                                          \
    BEGIN for (int i = 0; i < iters; ++i) { asm volatile(R8(OP8) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(k), "v"(m)); } \
    END out[blockIdx.x * blockDim.x + threadIdx.x] = (double)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);                 \
}
#define DEP64(NAME, T, INIT, OP1)                                                                                      \
__global__ void NAME(double* out, int iters) {                                                                          \
    T d0 = INIT;                                                                                                         \
    const T k = (T)1.0000001, m = (T)0.9999999;                                                                          \
    BEGIN for (int i = 0; i < iters; ++i) { asm volatile(R8(R8(OP1)) : "+v"(d0) : "v"(k), "v"(m)); }                    \
    END out[blockIdx.x * blockDim.x + threadIdx.x] = (double)d0;                                                        \
}
IND64(i_fma64, double, threadIdx.x, "v_fma_f64 %0, %0, %8, %9\nv_fma_f64 %1, %1, %8, %9\nv_fma_f64 %2, %2, %8, %9\nv_fma_f64 %3, %3, %8, %9\nv_fma_f64 %4, %4, %8, %9\nv_fma_f64 %5, %5, %8, %9\nv_fma_f64 %6, %6, %8, %9\nv_fma_f64 %7, %7, %8, %9\n")
DEP64(d_fma64, double, threadIdx.x, "v_fma_f64 %0, %0, %1, %2\n")
IND64(i_fma32, float, threadIdx.x + 1.5f, "v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\nv_fma_f32 %3, %3, %8, %9\nv_fma_f32 %4, %4, %8, %9\nv_fma_f32 %5, %5, %8, %9\nv_fma_f32 %6, %6, %8, %9\nv_fma_f32 %7, %7, %8, %9\n")
DEP64(d_fma32, float, threadIdx.x + 1.5f, "v_fma_f32 %0, %0, %1, %2\n")
IND64(i_pk, double, threadIdx.x, "v_pk_fma_f32 %0, %0, %8, %9\nv_pk_fma_f32 %1, %1, %8, %9\nv_pk_fma_f32 %2, %2, %8, %9\nv_pk_fma_f32 %3, %3, %8, %9\nv_pk_fma_f32 %4, %4, %8, %9\nv_pk_fma_f32 %5, %5, %8, %9\nv_pk_fma_f32 %6, %6, %8, %9\nv_pk_fma_f32 %7, %7, %8, %9\n")
DEP64(d_pk, double, threadIdx.x, "v_pk_fma_f32 %0, %0, %1, %2\n")
IND64(i_exp, float, threadIdx.x + 1.5f, "v_exp_f32 %0, %0\nv_exp_f32 %1, %1\nv_exp_f32 %2, %2\nv_exp_f32 %3, %3\nv_exp_f32 %4, %4\nv_exp_f32 %5, %5\nv_exp_f32 %6, %6\nv_exp_f32 %7, %7\n")
DEP64(d_exp, float, threadIdx.x + 1.5f, "v_exp_f32 %0, %0\n")
// a transcendental feeding an fp32 op feeding a transcendental ... (Box-Muller's log -> mul -> sqrt -> mul shape)
DEP64(d_exp_mul, float, threadIdx.x + 1.5f, "v_exp_f32 %0, %0\nv_mul_f32 %0, %0, %2\n")
__global__ void i_mad(double* out, int iters) {
    unsigned long long d0 = threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;
    const unsigned k = 0xD2511F53u, m = 0xCD9E8D57u + threadIdx.x;
    BEGIN for (int i = 0; i < iters; ++i) {
        asm volatile(R8("v_mad_u64_u32 %0, vcc, %8, %9, %0\nv_mad_u64_u32 %1, vcc, %8, %9, %1\nv_mad_u64_u32 %2, vcc, %8, %9, %2\nv_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                        "v_mad_u64_u32 %4, vcc, %8, %9, %4\nv_mad_u64_u32 %5, vcc, %8, %9, %5\nv_mad_u64_u32 %6, vcc, %8, %9, %6\nv_mad_u64_u32 %7, vcc, %8, %9, %7\n")
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(k), "v"(m) : "vcc");
    }
    END out[blockIdx.x * blockDim.x + threadIdx.x] = (double)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
}
__global__ void d_mad(double* out, int iters) {
    unsigned long long d0 = threadIdx.x;
    const unsigned k = 0xD2511F53u, m = 0xCD9E8D57u + threadIdx.x;
    BEGIN for (int i = 0; i < iters; ++i) { asm volatile(R8(R8("v_mad_u64_u32 %0, vcc, %1, %2, %0\n")) : "+v"(d0) : "v"(k), "v"(m) : "vcc"); }
    END out[blockIdx.x * blockDim.x + threadIdx.x] = (double)d0;
}
// LDS: a read and the wait for it between dependent fp64 work (the rollout's per-step row fetch)
__global__ void d_lds(double* out, int iters) {
    __shared__ double sh[512];
    sh[threadIdx.x] = threadIdx.x; sh[256 + threadIdx.x] = 1.0;
    __syncthreads();
    double d0 = threadIdx.x;
    int idx = threadIdx.x & 255;
    BEGIN for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 64; ++j) { d0 = d0 * 0.999 + sh[(idx + j) & 511]; }
    }
    END out[blockIdx.x * blockDim.x + threadIdx.x] = d0;
}
typedef void (*kern_t)(double*, int);
struct Case { const char* name; kern_t k; };
int main() {
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    double* out; CHK(hipMalloc(&out, sizeof(double) * cus * 8 * 256));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    std::vector<Case> cases = {{"fma_f64 independent x8", i_fma64}, {"fma_f64 DEPENDENT", d_fma64}, {"fma_f32 independent x8", i_fma32}, {"fma_f32 DEPENDENT", d_fma32},
                               {"pk_fma_f32 independent x8", i_pk}, {"pk_fma_f32 DEPENDENT", d_pk}, {"exp_f32 independent x8", i_exp}, {"exp_f32 DEPENDENT", d_exp},
                               {"exp->mul DEPENDENT pairs", d_exp_mul}, {"mad_u64_u32 independent x8", i_mad}, {"mad_u64_u32 DEPENDENT", d_mad},
                               {"fma_f64 <- ds_read DEPENDENT (64 per iter)", d_lds}};
    int wall_khz = 100000;
    hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    printf("device %s, %d CUs; cycles per instruction AS ONE WAVE SEES THEM (its own s_memtime), launch ms\n", prop.name, cus);
    for (int wps : {1, 2, 4}) {
        for (auto& c : cases) {
            const int iters = 2000, blocks = cus * wps;
            hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), 0, 0, out, 10);
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(a, 0));
            hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), 0, 0, out, iters);
            CHK(hipEventRecord(b, 0));
            CHK(hipEventSynchronize(b));
            float ms; CHK(hipEventElapsedTime(&ms, a, b));
            unsigned long long clk[2] = {0, 0};
            CHK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk)));
            const double mhz = clk[1] ? (double)clk[0] / (double)clk[1] * wall_khz * 1e-3 : 0.0;
            printf("wps=%d %-44s %7.3f ms  wave: %6.2f cycles/inst  (clock %.0f MHz; SIMD: %.2f cycles/inst)\n", wps, c.name, ms,
                   (double)clk[0] / (iters * 64.0), mhz, ms * 1e-3 * mhz * 1e6 / (iters * 64.0 * wps));
        }
    }
    return 0;
}
