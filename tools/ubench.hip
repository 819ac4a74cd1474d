// tools/ubench.hip -- per-instruction VALU issue cost on gfx950 (cycles per wave64 instruction per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o build/ubench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// shader clock under each kernel's own load: thread 0 of block 0 reads the shader cycle counter (s_memtime) and the
// constant-rate counter (s_memrealtime, hipDeviceAttributeWallClockRate) around its loop
__device__ unsigned long long g_clk[2];
#define CLK_BEGIN unsigned long long c0__ = clock64(), w0__ = wall_clock64();
#define CLK_END if (blockIdx.x == 0 && threadIdx.x == 0) { g_clk[0] = clock64() - c0__; g_clk[1] = wall_clock64() - w0__; }
#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
// 8 independent chains, 64 instructions per loop iteration
#define KERNEL64(NAME, ASM_D, ASM_S)                                                          \
__global__ void NAME(double* out, int iters) {                                                \
    double d0 = threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7; \
    const double k = 1.0000001, m = 0.9999999;                                                \
    CLK_BEGIN for (int i = 0; i < iters; ++i) {                                                         \
        for (int j = 0; j < 8; ++j) {                                                         \
            asm volatile(ASM_D : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(k), "v"(m)); \
        }                                                                                     \
    }                                                                                         \
    CLK_END out[blockIdx.x * blockDim.x + threadIdx.x] = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;       \
}
#define KERNEL32(NAME, ASM_D)                                                                 \
__global__ void NAME(double* out, int iters) {                                                \
    float d0 = threadIdx.x + 1.5f, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7; \
    const float k = 1.0000001f, m = 0.9999999f;                                               \
    CLK_BEGIN for (int i = 0; i < iters; ++i) {                                                         \
        for (int j = 0; j < 8; ++j) {                                                         \
            asm volatile(ASM_D : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(k), "v"(m)); \
        }                                                                                     \
    }                                                                                         \
    CLK_END out[blockIdx.x * blockDim.x + threadIdx.x] = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;       \
}
#define S8(fmt) fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7)

KERNEL64(k_fma_f64, "v_fma_f64 %0, %0, %8, %9\nv_fma_f64 %1, %1, %8, %9\nv_fma_f64 %2, %2, %8, %9\nv_fma_f64 %3, %3, %8, %9\nv_fma_f64 %4, %4, %8, %9\nv_fma_f64 %5, %5, %8, %9\nv_fma_f64 %6, %6, %8, %9\nv_fma_f64 %7, %7, %8, %9", "")
KERNEL64(k_mul_f64, "v_mul_f64 %0, %0, %8\nv_mul_f64 %1, %1, %8\nv_mul_f64 %2, %2, %8\nv_mul_f64 %3, %3, %8\nv_mul_f64 %4, %4, %8\nv_mul_f64 %5, %5, %8\nv_mul_f64 %6, %6, %8\nv_mul_f64 %7, %7, %8", "")
KERNEL64(k_add_f64, "v_add_f64 %0, %0, %8\nv_add_f64 %1, %1, %8\nv_add_f64 %2, %2, %8\nv_add_f64 %3, %3, %8\nv_add_f64 %4, %4, %8\nv_add_f64 %5, %5, %8\nv_add_f64 %6, %6, %8\nv_add_f64 %7, %7, %8", "")
KERNEL64(k_max_f64, "v_max_f64 %0, %0, %8\nv_max_f64 %1, %1, %8\nv_max_f64 %2, %2, %8\nv_max_f64 %3, %3, %8\nv_max_f64 %4, %4, %8\nv_max_f64 %5, %5, %8\nv_max_f64 %6, %6, %8\nv_max_f64 %7, %7, %8", "")
KERNEL32(k_fma_f32, "v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\nv_fma_f32 %3, %3, %8, %9\nv_fma_f32 %4, %4, %8, %9\nv_fma_f32 %5, %5, %8, %9\nv_fma_f32 %6, %6, %8, %9\nv_fma_f32 %7, %7, %8, %9")
KERNEL32(k_exp_f32, "v_exp_f32 %0, %0\nv_exp_f32 %1, %1\nv_exp_f32 %2, %2\nv_exp_f32 %3, %3\nv_exp_f32 %4, %4\nv_exp_f32 %5, %5\nv_exp_f32 %6, %6\nv_exp_f32 %7, %7")
KERNEL32(k_log_f32, "v_log_f32 %0, %0\nv_log_f32 %1, %1\nv_log_f32 %2, %2\nv_log_f32 %3, %3\nv_log_f32 %4, %4\nv_log_f32 %5, %5\nv_log_f32 %6, %6\nv_log_f32 %7, %7")
KERNEL32(k_sin_f32, "v_sin_f32 %0, %0\nv_sin_f32 %1, %1\nv_sin_f32 %2, %2\nv_sin_f32 %3, %3\nv_sin_f32 %4, %4\nv_sin_f32 %5, %5\nv_sin_f32 %6, %6\nv_sin_f32 %7, %7")
KERNEL32(k_sqrt_f32, "v_sqrt_f32 %0, %0\nv_sqrt_f32 %1, %1\nv_sqrt_f32 %2, %2\nv_sqrt_f32 %3, %3\nv_sqrt_f32 %4, %4\nv_sqrt_f32 %5, %5\nv_sqrt_f32 %6, %6\nv_sqrt_f32 %7, %7")
KERNEL32(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %8\nv_mul_lo_u32 %1, %1, %8\nv_mul_lo_u32 %2, %2, %8\nv_mul_lo_u32 %3, %3, %8\nv_mul_lo_u32 %4, %4, %8\nv_mul_lo_u32 %5, %5, %8\nv_mul_lo_u32 %6, %6, %8\nv_mul_lo_u32 %7, %7, %8")
KERNEL32(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %8\nv_mul_hi_u32 %1, %1, %8\nv_mul_hi_u32 %2, %2, %8\nv_mul_hi_u32 %3, %3, %8\nv_mul_hi_u32 %4, %4, %8\nv_mul_hi_u32 %5, %5, %8\nv_mul_hi_u32 %6, %6, %8\nv_mul_hi_u32 %7, %7, %8")
KERNEL32(k_xor_b32, "v_xor_b32 %0, %0, %8\nv_xor_b32 %1, %1, %8\nv_xor_b32 %2, %2, %8\nv_xor_b32 %3, %3, %8\nv_xor_b32 %4, %4, %8\nv_xor_b32 %5, %5, %8\nv_xor_b32 %6, %6, %8\nv_xor_b32 %7, %7, %8")
KERNEL64(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %8, %9\nv_pk_fma_f32 %1, %1, %8, %9\nv_pk_fma_f32 %2, %2, %8, %9\nv_pk_fma_f32 %3, %3, %8, %9\nv_pk_fma_f32 %4, %4, %8, %9\nv_pk_fma_f32 %5, %5, %8, %9\nv_pk_fma_f32 %6, %6, %8, %9\nv_pk_fma_f32 %7, %7, %8, %9", "")
// packed / VOP2 fp32 forms the two-samples-per-lane rollout would use
#define PK8(OP) OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n" OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8"
KERNEL64(k_pk_mul_f32, PK8("v_pk_mul_f32"), "")
KERNEL64(k_pk_add_f32, PK8("v_pk_add_f32"), "")
KERNEL32(k_add_f32, PK8("v_add_f32"))
KERNEL32(k_mul_f32, PK8("v_mul_f32"))
KERNEL32(k_max_f32, PK8("v_max_f32"))
KERNEL32(k_fmac_f32, PK8("v_fmac_f32"))
KERNEL32(k_med3_f32, "v_med3_f32 %0, %0, %8, %9\nv_med3_f32 %1, %1, %8, %9\nv_med3_f32 %2, %2, %8, %9\nv_med3_f32 %3, %3, %8, %9\nv_med3_f32 %4, %4, %8, %9\nv_med3_f32 %5, %5, %8, %9\nv_med3_f32 %6, %6, %8, %9\nv_med3_f32 %7, %7, %8, %9")
KERNEL32(k_add_f32_dpp, "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL32(k_bitop3, "v_bitop3_b32 %0, %0, %8, %9 bitop3:0x96\nv_bitop3_b32 %1, %1, %8, %9 bitop3:0x96\nv_bitop3_b32 %2, %2, %8, %9 bitop3:0x96\nv_bitop3_b32 %3, %3, %8, %9 bitop3:0x96\nv_bitop3_b32 %4, %4, %8, %9 bitop3:0x96\nv_bitop3_b32 %5, %5, %8, %9 bitop3:0x96\nv_bitop3_b32 %6, %6, %8, %9 bitop3:0x96\nv_bitop3_b32 %7, %7, %8, %9 bitop3:0x96")
KERNEL32(k_cndmask, "v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\nv_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc")
__global__ void k_cvt_f32_f64(double* out, int iters) {
    float d0 = threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;
    double k = 1.5 + threadIdx.x;
    CLK_BEGIN for (int i = 0; i < iters; ++i) {
        for (int j = 0; j < 8; ++j) {
            asm volatile("v_cvt_f32_f64 %0, %8\nv_cvt_f32_f64 %1, %8\nv_cvt_f32_f64 %2, %8\nv_cvt_f32_f64 %3, %8\nv_cvt_f32_f64 %4, %8\nv_cvt_f32_f64 %5, %8\nv_cvt_f32_f64 %6, %8\nv_cvt_f32_f64 %7, %8"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(k));
        }
    }
    CLK_END out[blockIdx.x * blockDim.x + threadIdx.x] = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;
}
__global__ void k_mad_u64_u32(double* out, int iters) {
    unsigned long long d0 = threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;
    const unsigned k = 0xD2511F53u, m = 0xCD9E8D57u + threadIdx.x;
    CLK_BEGIN for (int i = 0; i < iters; ++i) {
        for (int j = 0; j < 8; ++j) {
            asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\nv_mad_u64_u32 %1, vcc, %8, %9, %1\nv_mad_u64_u32 %2, vcc, %8, %9, %2\nv_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                         "v_mad_u64_u32 %4, vcc, %8, %9, %4\nv_mad_u64_u32 %5, vcc, %8, %9, %5\nv_mad_u64_u32 %6, vcc, %8, %9, %6\nv_mad_u64_u32 %7, vcc, %8, %9, %7"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(k), "v"(m) : "vcc");
        }
    }
    CLK_END out[blockIdx.x * blockDim.x + threadIdx.x] = (double)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
}
__global__ void k_cvt_f64_f32(double* out, int iters) {
    double d0 = threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;
    float k = 1.5f + threadIdx.x;
    CLK_BEGIN for (int i = 0; i < iters; ++i) {
        for (int j = 0; j < 8; ++j) {
            asm volatile("v_cvt_f64_f32 %0, %8\nv_cvt_f64_f32 %1, %8\nv_cvt_f64_f32 %2, %8\nv_cvt_f64_f32 %3, %8\nv_cvt_f64_f32 %4, %8\nv_cvt_f64_f32 %5, %8\nv_cvt_f64_f32 %6, %8\nv_cvt_f64_f32 %7, %8"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(k));
        }
    }
    CLK_END out[blockIdx.x * blockDim.x + threadIdx.x] = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;
}

typedef void (*kern_t)(double*, int);
struct Case { const char* name; kern_t k; };

int main() {
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double mhz = prop.clockRate / 1000.0;
    double* out; CHK(hipMalloc(&out, sizeof(double) * cus * 8 * 256));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    std::vector<Case> cases = {
        {"v_fma_f64", k_fma_f64}, {"v_mul_f64", k_mul_f64}, {"v_add_f64", k_add_f64}, {"v_max_f64", k_max_f64},
        {"v_fma_f32", k_fma_f32}, {"v_pk_fma_f32", k_pk_fma_f32}, {"v_exp_f32", k_exp_f32}, {"v_log_f32", k_log_f32},
        {"v_sin_f32", k_sin_f32}, {"v_sqrt_f32", k_sqrt_f32}, {"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_hi_u32", k_mul_hi_u32},
        {"v_xor_b32", k_xor_b32}, {"v_mad_u64_u32", k_mad_u64_u32}, {"v_cvt_f64_f32", k_cvt_f64_f32},
        {"v_pk_mul_f32", k_pk_mul_f32}, {"v_pk_add_f32", k_pk_add_f32}, {"v_add_f32", k_add_f32}, {"v_mul_f32", k_mul_f32},
        {"v_max_f32", k_max_f32}, {"v_fmac_f32", k_fmac_f32}, {"v_med3_f32", k_med3_f32}, {"v_add_f32_dpp", k_add_f32_dpp},
        {"v_bitop3_b32", k_bitop3}, {"v_cndmask_b32", k_cndmask}, {"v_cvt_f32_f64", k_cvt_f32_f64}};
    int wall_khz = 100000;
    hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    printf("device %s, %d CUs, clockRate %.0f MHz, wall clock %d kHz\n", prop.name, cus, mhz, wall_khz);
    for (int wps : {2, 4}) {   // waves per SIMD
        for (auto& c : cases) {
            const int iters = 2000;
            const int blocks = cus * wps;  // 256-thread blocks: 4 waves = 1 per SIMD
            hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), 0, 0, out, 10);
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(a, 0));
            hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), 0, 0, out, iters);
            CHK(hipEventRecord(b, 0));
            CHK(hipEventSynchronize(b));
            float ms; CHK(hipEventElapsedTime(&ms, a, b));
            const double insts_per_simd = (double)iters * 64 * wps;
            unsigned long long clk[2] = {0, 0};
            CHK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk)));
            const double real_mhz = clk[1] ? (double)clk[0] / (double)clk[1] * wall_khz * 1e-3 : 0.0;
            printf("wps=%d %-14s %8.3f ms  -> %.2f ns/inst/SIMD = %.2f cycles @2.4GHz; measured shader clock %.0f MHz -> %.2f cycles/inst (block 0: %.2f cycles/inst by its own counter)\n",
                   wps, c.name, ms, ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4, real_mhz,
                   ms * 1e6 / insts_per_simd * real_mhz * 1e-3, (double)clk[0] / insts_per_simd);
        }
    }
    return 0;
}
