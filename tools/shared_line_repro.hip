// shared_line_repro.hip -- do two engines' kernels, in flight on two unordered streams, get each other's words of a SHARED
// 128-byte line wrong?  (VERDICT r5 item 3 / EXPERIMENTS.md 56-57.)
//
// The layout of round 5's measurement build that shared the per-wave noise sums: R rows of `pitch` 4-byte words (15 625 at config 4:
// NOT a multiple of a 32-word line), engine A owns words [0, cut) of every row, engine B words [cut, n) -- so the cut falls inside a
// line of (almost) every row.  Per "tick", on its own stream, each engine runs
//     W_X  writes f(tick, row, word) to its words  -- the rollout kernels' store: raw buffer store, aux = sc1 (write-through)
//     R_X  reads its words back with plain loads    -- the update kernel's read -- and counts the words that are not f(tick, ...)
// for `ticks` ticks, the two streams never synchronised with each other (the co-scheduled tick's pattern: one engine's R runs while
// the other's W writes its own words of the same lines through other XCDs' L2s).
// Variants (one line of output each):
//     shared    cut = 9088 + 7 words  : every row's cut inside a line                  -> stale words, if sharing a line is unsafe
//     aligned   pitch 15 648, cut 9088: rows and cut on line boundaries (round 6 layout) -> control
//     plain     shared, W with plain (write-back) stores instead of sc1
//     overrun   B's W also writes a 0 ONE word past its last own word of every row.  B's region ends at the row's end, so that word
//               is word 0 of the NEXT row -- engine A's.  (What the mixed rollout's zero fill of a wave's second 64-sample slot did in
//               the round-5 measurement build at K = 10^6: a shard whose last wave holds 64 samples, its rows columns of the handle's.)
// Build: hipcc --offload-arch=gfx950 -O2 tools/shared_line_repro.hip -o tools/shared_line_repro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned val(unsigned tick, unsigned row, unsigned w) { return tick * 2654435761u + row * 40503u + w * 7u + 1u; }

// one thread per word of the engine's region [w0, w1) of every row; `spin` adds ALU work so that a launch lives long enough to overlap
template <bool SC1>
__global__ void write_kernel(unsigned* buf, int rows, int pitch, int w0, int w1, unsigned tick, int spin, int overrun_at) {
    const int w = w0 + blockIdx.x * blockDim.x + threadIdx.x, row = blockIdx.y;
    float acc = (float)w;
    for (int i = 0; i < spin; ++i) acc = __builtin_fmaf(acc, 1.0000001f, 0.5f);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, (int)((size_t)rows * pitch * 4), 0x00020000);
    const bool mine = w < w1;
    const unsigned v = val(tick, row, w) + (acc == 12345.f ? 1u : 0u);
    if (SC1) __builtin_amdgcn_raw_buffer_store_b32(v, rs, mine ? (unsigned)(((size_t)row * pitch + w) * 4) : 0xFFFFFFFFu, 0, 16);
    else if (mine) buf[(size_t)row * pitch + w] = v;
    // the round-5 bug: a 0 one word past the region's last word -- at the row's end that is word 0 of the next row
    if (overrun_at >= 0 && w == w1 - 1 && row + 1 < rows) {
        const size_t at = (size_t)row * pitch + overrun_at;
        if (SC1) __builtin_amdgcn_raw_buffer_store_b32(0u, rs, (unsigned)(at * 4), 0, 16);
        else buf[at] = 0u;
    }
}
__global__ void read_kernel(const unsigned* buf, int rows, int pitch, int w0, int w1, unsigned tick, unsigned long long* bad, unsigned* first) {
    const int w = w0 + blockIdx.x * blockDim.x + threadIdx.x, row = blockIdx.y;
    if (w >= w1) return;
    const unsigned got = buf[(size_t)row * pitch + w], want = val(tick, row, w);
    if (got != want) {
        if (atomicAdd(bad, 1ull) == 0) { first[0] = tick; first[1] = (unsigned)row; first[2] = (unsigned)w; first[3] = got; first[4] = want; }
    }
}

struct Result { unsigned long long badA, badB; unsigned firstA[5], firstB[5]; float ms; };

Result run(int rows, int pitch, int n, int cut, bool sc1, int ticks, int spin, bool overrun) {
    unsigned* buf; unsigned long long* bad; unsigned* first;
    CK(hipMalloc(&buf, (size_t)rows * pitch * 4));
    CK(hipMemset(buf, 0, (size_t)rows * pitch * 4));
    CK(hipMalloc(&bad, 16)); CK(hipMemset(bad, 0, 16));
    CK(hipMalloc(&first, 40)); CK(hipMemset(first, 0, 40));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 ga((cut + 255) / 256, rows), gb((n - cut + 255) / 256, rows);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, sa));
    for (int t = 1; t <= ticks; ++t) {
        // engine A: 58 % of the samples, launched first -- as the co-scheduled tick does
        if (sc1) hipLaunchKernelGGL(write_kernel<true>, ga, dim3(256), 0, sa, buf, rows, pitch, 0, cut, (unsigned)t, spin, -1);
        else hipLaunchKernelGGL(write_kernel<false>, ga, dim3(256), 0, sa, buf, rows, pitch, 0, cut, (unsigned)t, spin, -1);
        // (B's region ends at the row's end: "one past" = pitch = word 0 of the next row)
        if (sc1) hipLaunchKernelGGL(write_kernel<true>, gb, dim3(256), 0, sb, buf, rows, pitch, cut, n, (unsigned)t, spin, overrun ? pitch : -1);
        else hipLaunchKernelGGL(write_kernel<false>, gb, dim3(256), 0, sb, buf, rows, pitch, cut, n, (unsigned)t, spin, overrun ? pitch : -1);
        hipLaunchKernelGGL(read_kernel, ga, dim3(256), 0, sa, buf, rows, pitch, 0, cut, (unsigned)t, bad, first);
        hipLaunchKernelGGL(read_kernel, gb, dim3(256), 0, sb, buf, rows, pitch, cut, n, (unsigned)t, bad + 1, first + 5);
    }
    CK(hipEventRecord(e1, sa));
    CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
    Result r{};
    unsigned long long h[2]; unsigned f[10];
    CK(hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(f, first, 40, hipMemcpyDeviceToHost));
    r.badA = h[0]; r.badB = h[1]; memcpy(r.firstA, f, 20); memcpy(r.firstB, f + 5, 20);
    CK(hipEventElapsedTime(&r.ms, e0, e1));
    CK(hipFree(buf)); CK(hipFree(bad)); CK(hipFree(first));
    CK(hipStreamDestroy(sa)); CK(hipStreamDestroy(sb));
    return r;
}

int main(int argc, char** argv) {
    const int ticks = argc > 1 ? atoi(argv[1]) : 3000, spin = argc > 2 ? atoi(argv[2]) : 200;
    const int rows = 100;   // config 4: T * 2 rows of per-wave sums
    struct V { const char* name; int pitch, n, cut; bool sc1, overrun; } vs[] = {
        {"shared  (cut inside a line of every row, sc1 stores, plain loads)", 15625, 15625, 9088 + 7, true, false},
        {"shared  (the round-5 build's own cut: 9088 words, rows of 15 625)  ", 15625, 15625, 9088, true, false},
        {"aligned (rows and cut on line boundaries: the round-6 layout)      ", 15648, 15625, 9088, true, false},
        {"plain   (shared lines, write-back stores)                          ", 15625, 15625, 9088 + 7, false, false},
        {"overrun (round-5 rows + the zero fill one slot past the shard)      ", 15625, 15625, 9088, true, true},
    };
    printf("# %d ticks per variant, two unordered streams, %d rows; words checked per tick: %d\n", ticks, rows, rows * 15625);
    for (const V& v : vs) {
        const Result r = run(rows, v.pitch, v.n, v.cut, v.sc1, ticks, spin, v.overrun);
        printf("%s  wrong words: A %llu  B %llu   (%.1f us per tick)", v.name, r.badA, r.badB, 1e3 * r.ms / ticks);
        if (r.badA) printf("   first A: tick %u row %u word %u got %u want %u", r.firstA[0], r.firstA[1], r.firstA[2], r.firstA[3], r.firstA[4]);
        if (r.badB) printf("   first B: tick %u row %u word %u got %u want %u", r.firstB[0], r.firstB[1], r.firstB[2], r.firstB[3], r.firstB[4]);
        printf("\n");
    }
    return 0;
}
