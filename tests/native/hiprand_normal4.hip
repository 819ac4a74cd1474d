// Test helper (GPU tests only): hipRAND's own normals,
//     hiprand_init(seed, subsequence, offset, &state);  hiprand_normal4(&state)
// for the (seed, subsequence, offset / 4) triples read from stdin, one per line -> the four floats' bit patterns per line on
// stdout.  tests/test_gpu_parity.py compares them BIT FOR BIT with what the engine draws under option "noise_packing" = 2
// (mppi_download_noise / sigma): that noise mode IS hiprand_normal4 on the engine's counters (subsequence = agent << 32 | tick,
// offset = 4 * (draw << 32 | global sample); the four values = steps 2 * draw, 2 * draw + 1 x wheels 0, 1).
#include <hip/hip_runtime.h>
#include <hiprand/hiprand_kernel.h>

#include <cstdio>
#include <cstring>
#include <vector>

__global__ void draw(const unsigned long long* in, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    hiprandStatePhilox4_32_10_t st;
    hiprand_init(in[3 * i], in[3 * i + 1], 4ull * in[3 * i + 2], &st);
    const float4 r = hiprand_normal4(&st);
    out[4 * i] = r.x; out[4 * i + 1] = r.y; out[4 * i + 2] = r.z; out[4 * i + 3] = r.w;
}

int main() {
    std::vector<unsigned long long> in;
    unsigned long long a, b, c;
    while (std::scanf("%llu %llu %llu", &a, &b, &c) == 3) { in.push_back(a); in.push_back(b); in.push_back(c); }
    const int n = (int)(in.size() / 3);
    if (n == 0) return 2;
    unsigned long long* d_in = nullptr;
    float* d_out = nullptr;
    if (hipMalloc(&d_in, in.size() * sizeof(unsigned long long)) != hipSuccess || hipMalloc(&d_out, (size_t)n * 16) != hipSuccess) return 3;
    if (hipMemcpy(d_in, in.data(), in.size() * sizeof(unsigned long long), hipMemcpyHostToDevice) != hipSuccess) return 3;
    hipLaunchKernelGGL(draw, dim3((n + 63) / 64), dim3(64), 0, 0, d_in, d_out, n);
    std::vector<float> out((size_t)n * 4);
    if (hipMemcpy(out.data(), d_out, (size_t)n * 16, hipMemcpyDeviceToHost) != hipSuccess) return 4;
    for (int i = 0; i < n; ++i) {
        unsigned w[4];
        std::memcpy(w, &out[4 * i], 16);
        std::printf("%u %u %u %u\n", w[0], w[1], w[2], w[3]);
    }
    return 0;
}
