// rollout_fused.hpp -- rollout_fused_kernel: rollout + cost-to-go + softmax partials of a lane-per-sample tick in ONE kernel, for
// the reference's own precision (MPPI_STORE_F64: the reference is float64 end to end, control/src/mppi:127-208).  The two-kernel
// tick of that mode is memory-bound twice over -- the rollout writes 8 B per sample-step (400 MB per tick at config 4, more than
// the 256 MB Infinity Cache holds), the update reads them back -- and its two big launches run one after the other (110 + 70 us).
// Here V never leaves the chip (EXPERIMENTS.md 46: the estimate; 58: the build, round 6):
//
//   one WAVE (four to a workgroup: one per SIMD of its CU, each on its own) walks groups of 64 samples (lane = sample), T
//   sequential steps each -- the all-fp64 lean step of rollout_kernel, operation for operation -- and drops the running cost
//   prefix of every step into LDS, pf[t][lane];
//   behind the loop the wave turns round: lane = ROW t, and folds its 64 samples into the row's softmax tuple
//       v_k = Stot_k - pf[t][k]        (V[t][k] = base[t] + v_k: total minus exclusive prefix, control/src/mppi:175)
//       m = min_k v_k;   e_k = exp2((M - v_k) log2e / lambda);   D = sum e_k,   N = sum e_k eps_k[t]       (:189-196)
//   in ONE pass over the wave's own LDS: the group's minimum, and -- as bits per lane -- the samples that can still carry weight
//   against the wave's RUNNING minimum M (almost none after the first groups); only those are visited: their weight in fp64, their
//   noise re-drawn from its Philox counter (lane t draws (sample k, step t): one call serves every row that needs it);
//   the group's tuple is merged into the wave's running one (exact rescaling), the four waves of a workgroup merge theirs, and
//   the workgroup leaves ONE tuple per row: part[a][t][workgroup] -- 256 tuples per row for the merge launch, whatever K is.
//
// LDS per wave: the prefix [T][66] doubles (rows 16-byte aligned: read as pairs; a lane group's reads land in distinct banks), 64
// totals, the eps sums -- 27 KB at T = 50; a workgroup = four waves + the per-step table = 110 KB: one workgroup per CU, one wave
// per SIMD, 1024 waves on the chip, each walking ceil(K / 65536) groups (static, strided: the same wave folds the same samples in
// the same order every run).  The per-step table of the nominal trajectory is LOADED (the previous tick's finalize kernel or
// nominal_kernel computed it): no prologue per wave; a chunk's rows are requested at its top, in front of its Philox draws (a lone
// wave per SIMD has nobody to hide an LDS round trip behind).
// SPLIT (T <= 56: what the CU's LDS holds next to the prefix rows): a workgroup of EIGHT waves -- every walking wave gets a DRAWING wave
// on its SIMD (the hardware deals a workgroup's waves to the four SIMDs in turn) that makes its noise (Philox, Box-Muller, the eps sums: a
// third of a chunk's issue cycles) up to two chunks ahead and hands it over through two LDS buffers and a pair of counters: a lone
// wave leaves a third of its SIMD's issue slots empty, and the drawing wave lives in them -- config 4 176 -> 162 us per tick
// (EXPERIMENTS.md 67).  Same functions, same operands, the same groups in the same order: bit-identical tuples.
// Serves: fp64 storage, device noise not stored, rk4 + dd_dynamics, Q = diag(q, q, 0), no obstacle grid, T <= 64, the default
// noise stream.  Far from the goal a group has a handful of (row, sample) pairs with weight; parked AT the goal a few per cent of
// all pairs carry weight and every one costs a Philox call here -- the engine keeps the two-kernel tick for that regime (it reads
// the last tick's largest row sum of weights from the pinned outputs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mppi_kernels.hpp"

namespace mppi {

constexpr int kFusedPitch = 66;   // doubles per prefix row in LDS (64 samples + 2: rows stay 16-byte aligned and a row pair's 16-byte reads of one lane group land in distinct banks)
constexpr int kFusedRow = 6;      // doubles per table row in LDS ({un0, un1, w0, w1, cb, -}: three 16-byte reads)
// LDS bytes of one wave (prefix [T][66] f64, totals [64] f64, eps sums [T][2] f32, rounded to 16) and of a workgroup (table [T][6] f64 + four waves)
// (split form: two sets of eps sums per wave -- the drawing wave is a group ahead of the fold -- and the pairs' noise buffers)
constexpr int kFusedNBuf = 2;     // split form: noise buffers per wave pair (the drawing wave runs up to two chunks ahead)
constexpr int kFusedBufSteps = 6 + 3;   // steps a buffer holds: a chunk + the steps that ride along with the last one
inline size_t rollout_fused_lds_wave(int T, bool split = false) { return ((size_t)T * kFusedPitch * 8 + 64 * 8 + (size_t)T * 2 * 4 * (split ? 2 : 1) + 15) / 16 * 16; }
inline size_t rollout_fused_lds(int T, bool split = false) {
    return ((size_t)T * kFusedRow * 8 + 15) / 16 * 16 + 4 * rollout_fused_lds_wave(T, split) + (split ? (size_t)4 * kFusedNBuf * kFusedBufSteps * 64 * 8 : 0);
}

struct RolloutFusedArgs {
    DevParams P;
    hipStream_t stream;
    uint64_t seed;
    uint32_t tick;
    const uint32_t* tick_ptr;
    const double *state, *goal, *unom;
    const double* tc;     // [A][T][8] the nominal trajectory's per-step table (rows {un0, un1, w0, w1, cb}; row 0 also (cos, sin) of the pose's heading)
    double* part;         // [A][T][NB / 4][8]: one tuple per row and workgroup
    int NB;               // waves per agent (a multiple of 4: four to a workgroup)
    int nterm;            // 4 | 7
    bool split;           // eight waves per workgroup: waves 4-7 draw the noise for waves 0-3 (rollout_fused_kernel, SPLIT)
};
hipError_t launch_rollout_fused(const RolloutFusedArgs& a);

#ifdef MPPI_ROLLOUT_FUSED_TU
// SPLIT: a workgroup of EIGHT waves -- waves 4-7 draw the noise (Philox, Box-Muller, the per-wave eps sums: a third of a chunk's
// issue cycles) for waves 0-3, up to two chunks ahead, through LDS buffers and a pair of counters per wave pair; waves 0-3 walk the
// dynamics and fold as ever.  The hardware deals a workgroup's waves to the SIMDs in turn (wave i + 4 next to wave i:
// tools/simd_map.hip): every SIMD hosts a walking wave and, in the issue slots that lone wave leaves empty, the drawing wave that
// feeds it (EXPERIMENTS.md 66, 67).  Same functions, same operands, the same groups in the same order: bit-identical tuples.
template <int NTERM, bool SPLIT>
__global__ __launch_bounds__(SPLIT ? 512 : 256) __attribute__((amdgpu_waves_per_eu(SPLIT ? 2 : 1, SPLIT ? 2 : 1))) void rollout_fused_kernel(
    DevParams P, const double* __restrict__ state, const double* __restrict__ goal, const double* __restrict__ unom,
    const double* __restrict__ tc, uint64_t seed, uint32_t tick_arg, const uint32_t* __restrict__ tick_ptr, double* __restrict__ part, int NB) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int T = P.T, tid = threadIdx.x, lane = tid & 63, wid = (tid >> 6) & 3, a = blockIdx.y, wv = (int)blockIdx.x * 4 + wid;
    const bool drawer = SPLIT && __builtin_amdgcn_readfirstlane(tid >> 8) != 0;   // (wave-uniform)
    const size_t wave_bytes = ((size_t)T * kFusedPitch * 8 + 64 * 8 + (size_t)T * 2 * 4 * (SPLIT ? 2 : 1) + 15) / 16 * 16;
    double* lt = reinterpret_cast<double*>(smem_raw);          // [T][kFusedRow]   (the workgroup's)
    char* mine = smem_raw + ((size_t)T * kFusedRow * 8 + 15) / 16 * 16 + (size_t)wid * wave_bytes;
    double* pf = reinterpret_cast<double*>(mine);              // [T][kFusedPitch]   (this wave's -- SPLIT: this wave pair's --, like everything below)
    double* st = pf + (size_t)T * kFusedPitch;                 // [64]
    float* es0 = reinterpret_cast<float*>(st + 64);            // [T][2]  (SPLIT: [2][T][2], by the parity of the pair's group count)
    // SPLIT: the pair's noise buffers [kFusedNBuf][kFusedBufSteps][64] of {wheel 0, wheel 1}, and its hand-over words: chunks drawn (the
    // drawing wave's word), chunks taken (the walking wave's) -- each set to 0 by its writer in front of the first barrier
    typedef float f2n __attribute__((ext_vector_type(2)));
    f2n* const nzb = reinterpret_cast<f2n*>(smem_raw + ((size_t)T * kFusedRow * 8 + 15) / 16 * 16 + 4 * wave_bytes) + (size_t)wid * kFusedNBuf * kFusedBufSteps * 64 + lane;
    __shared__ int ho_drawn[SPLIT ? 4 : 1], ho_taken[SPLIT ? 4 : 1];
    if (SPLIT && lane == 0) { if (drawer) ho_drawn[wid] = 0; else ho_taken[wid] = 0; }
    // this wave's own LDS traffic needs no workgroup barrier: a wave's LDS operations complete in order; the fences keep the compiler in line
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    ClockProbe probe(P);
    snapshot_inputs(P, state, goal, unom, a);
    const double st_x = state[a * 3 + 0], st_y = state[a * 3 + 1], st_th = state[a * 3 + 2];
    const double g_x = goal[a * 3 + 0], g_y = goal[a * 3 + 1], g_th = goal[a * 3 + 2];
    const uint32_t tick = tick_ptr ? *tick_ptr : tick_arg;
    const double half_kd = 0.5 * P.kth * P.dt;
    for (int i = tid; i < T * kFusedRow; i += (SPLIT ? 512 : 256)) {
        const double v = tc[((size_t)a * T + i / kFusedRow) * kTcW + i % kFusedRow];   // (word 5 of a row: not used)
        lt[i] = (i % kFusedRow < 2) ? v * half_kd : v;
    }
    const double head_c = tc[(size_t)a * T * kTcW + 5], head_s = tc[(size_t)a * T * kTcW + 6];
    __syncthreads();
    const double p_max = half_kd * P.u_max, f = P.lean_f, rho = P.lean_rho, inv_sq = P.lean_inv_f;
    const uint32_t key0 = (uint32_t)seed, key1 = (uint32_t)(seed >> 32);
    const float sigf = (float)P.sigma;
    const double scale = P.inv_lambda * 1.4426950408889634;   // log2(e) / lambda
    const double cut = -100.0;                                // log2 of the smallest weight that is kept (the update kernel's fp64 cut)
    const int groups = (P.K + 63) >> 6;
    const int trow = lane < T ? lane : T - 1;                 // this lane's ROW in the folding pass (lanes >= T shadow the last row)

    constexpr int U = 6;
    const int T4 = T - T % U;
    const bool ride = T4 >= U && (T - T4 == 1 || T - T4 == 2);   // (uniform) as rollout_kernel: the one or two steps behind the last full chunk
    // a group's chunks as a list of (first step, whether a tail, whether the steps behind the last full chunk ride along): what a
    // walking wave integrates and -- SPLIT -- what its drawing wave hands over, one entry at a time
    auto for_each_chunk = [&](auto&& fn) __attribute__((always_inline)) {
        const int t_loop = ride ? T4 - U : T4;
        for (int t0 = 0; t0 < t_loop; t0 += U) fn(t0, false, std::false_type{});
        if (ride) fn(T4 - U, false, std::true_type{});
        else if (T4 < T) fn(T4, true, std::false_type{});
    };
    if constexpr (SPLIT) {
        if (drawer) {
            int ev = 0, gi = 0;
            for (int g = wv; g < groups; g += NB, ++gi) {
                const int k = g * 64 + lane;
                const bool active = k < P.K;
                const uint32_t ctr0 = P.sample_offset + (uint32_t)k;
                float* const es = es0 + (size_t)(gi & 1) * T * 2;
                for_each_chunk([&](int t0, bool tail, auto extra_tag) __attribute__((always_inline)) {
                    constexpr bool EXTRA = decltype(extra_tag)::value;
                    float nz[U][2], tz[kStepsPerDraw][2];
#pragma unroll
                    for (int j = 0; j < U; j += kStepsPerDraw) {
                        if (!tail || t0 + j < T) {
                            float e[6];
                            philox_normals(ctr0, (uint32_t)((t0 + j) / kStepsPerDraw), tick, P.agent_offset + (uint32_t)a, key0, key1, sigf, e);
#pragma unroll
                            for (int i = 0; i < kStepsPerDraw; ++i) { nz[j + i][0] = e[2 * i]; nz[j + i][1] = e[2 * i + 1]; }
                        } else {
#pragma unroll
                            for (int i = 0; i < kStepsPerDraw; ++i) { nz[j + i][0] = 0.f; nz[j + i][1] = 0.f; }
                        }
                    }
                    if constexpr (EXTRA) {
                        float e[6];
                        philox_normals(ctr0, (uint32_t)(T4 / kStepsPerDraw), tick, P.agent_offset + (uint32_t)a, key0, key1, sigf, e);
#pragma unroll
                        for (int i = 0; i < kStepsPerDraw; ++i) { tz[i][0] = T4 + i < T ? e[2 * i] : 0.f; tz[i][1] = T4 + i < T ? e[2 * i + 1] : 0.f; }
                    }
                    {   // per-wave sums of eps (the walking wave's eps_sums, on the floats it would convert back)
                        float sv[16];
#pragma unroll
                        for (int j = 0; j < U; ++j) { sv[2 * j] = active ? nz[j][0] : 0.f; sv[2 * j + 1] = active ? nz[j][1] : 0.f; }
#pragma unroll
                        for (int j = 2 * U; j < 16; ++j) sv[j] = !EXTRA ? 0.f : (active ? tz[(j - 2 * U) >> 1][j & 1] : 0.f);
                        const float tot = wave_sum16<EXTRA>(sv, lane);
                        const int idx = sum16_index(lane), te = t0 + (idx >> 1);
                        if (lane < 16 && idx < (EXTRA ? 16 : 2 * U) && te < T) es[te * 2 + (idx & 1)] = tot;
                    }
                    // the buffer is free once the chunk drawn into it kFusedNBuf hand-overs ago has been taken
                    if (ev >= kFusedNBuf)
                        while (__hip_atomic_load(&ho_taken[wid], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < ev - kFusedNBuf + 1) __builtin_amdgcn_s_sleep(1);
                    f2n* const buf = nzb + (size_t)(ev % kFusedNBuf) * kFusedBufSteps * 64;
#pragma unroll
                    for (int j = 0; j < U; ++j) buf[j * 64] = f2n{nz[j][0], nz[j][1]};
                    if constexpr (EXTRA) {
#pragma unroll
                        for (int j = 0; j < kStepsPerDraw; ++j) buf[(U + j) * 64] = f2n{tz[j][0], tz[j][1]};
                    }
                    ++ev;
                    __hip_atomic_store(&ho_drawn[wid], ev, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);   // (every lane stores the same count)
                });
            }
            __syncthreads();   // (the barrier in front of the walking waves' merge)
            return;
        }
    }
    int ev = 0, gi = 0;   // (SPLIT) chunks taken, groups begun

    // the wave's running tuple of row `lane`
    double Mr = INFINITY, Dr = 0.0, N0r = 0.0, N1r = 0.0, E0r = 0.0, E1r = 0.0, Cr = 0.0;

    for (int g = wv; g < groups; g += NB, ++gi) {
        const int k = g * 64 + lane;
        const bool active = k < P.K;
        const uint32_t ctr0 = P.sample_offset + (uint32_t)k;
        float* const es = es0 + (SPLIT ? (size_t)(gi & 1) * T * 2 : 0);
        double x = (st_x - g_x) * f, y = (st_y - g_y) * f, th = st_th;
        double c = head_c * rho, s = head_s * rho;
        double pre = 0.0;
        double cur[U][2], tl[kStepsPerDraw][2];
        auto draw_chunk = [&](int t0, bool tail) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < U; j += kStepsPerDraw) {
                if (!tail || t0 + j < T) {
                    float e[6];
                    philox_normals(ctr0, (uint32_t)((t0 + j) / kStepsPerDraw), tick, P.agent_offset + (uint32_t)a, key0, key1, sigf, e);
#pragma unroll
                    for (int i = 0; i < kStepsPerDraw; ++i) { cur[j + i][0] = (double)e[2 * i]; cur[j + i][1] = (double)e[2 * i + 1]; }
                } else {
#pragma unroll
                    for (int i = 0; i < kStepsPerDraw; ++i) { cur[j + i][0] = 0.0; cur[j + i][1] = 0.0; }
                }
            }
        };
        // per-wave sums of eps for the chunk's steps (the E of the softmax floor term, control/src/mppi:193): fp32, as rollout_kernel forms them
        auto eps_sums = [&](int t0, auto extra_tag) __attribute__((always_inline)) {
            constexpr bool EXTRA = decltype(extra_tag)::value;
            float ev[16];
#pragma unroll
            for (int j = 0; j < U; ++j) { ev[2 * j] = active ? (float)cur[j][0] : 0.f; ev[2 * j + 1] = active ? (float)cur[j][1] : 0.f; }
#pragma unroll
            for (int j = 2 * U; j < 16; ++j) ev[j] = !EXTRA ? 0.f : (active ? (float)tl[(j - 2 * U) >> 1][j & 1] : 0.f);
            const float tot = wave_sum16<EXTRA>(ev, lane);
            const int idx = sum16_index(lane), te = t0 + (idx >> 1);
            if (lane < 16 && idx < (EXTRA ? 16 : 2 * U) && te < T) es[te * 2 + (idx & 1)] = tot;
        };
        // one step: the prefix BEFORE the step goes to LDS, then explore + clip + rk4 + cost (rollout_kernel's lean step, operation for operation)
        // the chunk's table rows, requested at the top of the chunk (in front of its Philox draws): a lone wave per SIMD has nobody to
        // hide an LDS round trip per step behind
        typedef double d2 __attribute__((ext_vector_type(2)));
        d2 tb[U][3];
        auto load_rows = [&](int t0, int n) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < U; ++j)
                if (j < n) {
                    const d2* r = reinterpret_cast<const d2*>(lt + (t0 + j) * kFusedRow);
                    tb[j][0] = r[0]; tb[j][1] = r[1]; tb[j][2] = r[2];
                }
        };
        auto step = [&](int t, int j, double e0, double e1) __attribute__((always_inline)) {
            const double un0 = tb[j][0].x, un1 = tb[j][0].y, w0 = tb[j][1].x, w1 = tb[j][1].y, cb = tb[j][2].x;
            pf[t * kFusedPitch + lane] = pre;
            const double p0 = clamp_sym(fma(e0, half_kd, un0), p_max), p1 = clamp_sym(fma(e1, half_kd, un1), p_max);
            const double phi = p1 - p0;
            double sp, cp;
            small_sincos<NTERM>(phi, sp, cp);
            const double c1 = c * cp - s * sp, s1 = s * cp + c * sp;
            const double tcp2 = twice(cp);
            const double c2 = fma(tcp2, c1, -c), s2 = fma(tcp2, s1, -s);
            const double gg = (p0 + p1) * (tcp2 + 4.0);
            x = fma(gg, c1, x);
            y = fma(gg, s1, y);
            th = fma(2.0, phi, th);
            c = c2; s = s2;
            double dc = fma(x, x, fma(y, y, cb));
            dc = fma(w0, e0, dc);
            dc = fma(w1, e1, dc);
            pre += dc;
        };
        auto integrate = [&](int t0, bool guard) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < U; ++j)
                if (!guard || t0 + j < T) step(t0 + j, j, cur[j][0], cur[j][1]);
        };
        // SPLIT: the chunk's noise from the pair's buffer (the floats the drawing wave left; widened here as draw_chunk widens them)
        auto take_chunk = [&](auto extra_tag) __attribute__((always_inline)) {
            constexpr bool EXTRA = decltype(extra_tag)::value;
            const f2n* const buf = nzb + (size_t)(ev % kFusedNBuf) * kFusedBufSteps * 64;
            ++ev;
            while (__hip_atomic_load(&ho_drawn[wid], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < ev) __builtin_amdgcn_s_sleep(1);
#pragma unroll
            for (int j = 0; j < U; ++j) { const f2n v = buf[j * 64]; cur[j][0] = (double)v.x; cur[j][1] = (double)v.y; }
            if constexpr (EXTRA) {
#pragma unroll
                for (int j = 0; j < kStepsPerDraw; ++j) { const f2n v = buf[(U + j) * 64]; tl[j][0] = (double)v.x; tl[j][1] = (double)v.y; }
            }
            // (release: the reads above are complete before the count that frees the buffer is visible)
            __hip_atomic_store(&ho_taken[wid], ev, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        for_each_chunk([&](int t0, bool tail, auto extra_tag) __attribute__((always_inline)) {
            constexpr bool EXTRA = decltype(extra_tag)::value;
            load_rows(t0, tail ? T - T4 : U);
            if constexpr (SPLIT) take_chunk(extra_tag);
            else {
                draw_chunk(t0, tail);
                if constexpr (EXTRA) {
                    float e[6];
                    philox_normals(ctr0, (uint32_t)(T4 / kStepsPerDraw), tick, P.agent_offset + (uint32_t)a, key0, key1, sigf, e);
#pragma unroll
                    for (int i = 0; i < kStepsPerDraw; ++i) {
                        tl[i][0] = T4 + i < T ? (double)e[2 * i] : 0.0;
                        tl[i][1] = T4 + i < T ? (double)e[2 * i + 1] : 0.0;
                    }
                }
                eps_sums(t0, extra_tag);
            }
            integrate(t0, tail);
            if constexpr (EXTRA) {
#pragma unroll
                for (int j = 0; j < U; ++j) { cur[j][0] = j < kStepsPerDraw ? tl[j][0] : 0.0; cur[j][1] = j < kStepsPerDraw ? tl[j][1] : 0.0; }
                load_rows(T4, 2);
                integrate(T4, true);
            }
        });
        {   // terminal cost (control/src/mppi:165-173); the theta error is not wrapped beyond rk4's own wrap
            const double thw = (th > M_PI || th <= -M_PI) ? wrap_theta(th) : th;
            const double dx = x * inv_sq, dy = y * inv_sq, dth = thw - g_th;
            pre += P.p0 * dx * dx + P.p1 * dy * dy + P.p2 * dth * dth;
        }
        st[lane] = active ? pre : INFINITY;
        wave_sync();

        // ---- lanes = rows: fold the group's 64 samples into row `trow`'s tuple ------------------------------------------------------
        const d2* prow2 = reinterpret_cast<const d2*>(pf + (size_t)trow * kFusedPitch);
        const d2* st2 = reinterpret_cast<const d2*>(st);
        // ONE pass over the row's 64 values for both the group's minimum and the samples that can matter: a weight is formed relative to
        // the minimum the wave's running tuple will have BEHIND this group, Mn = min(Mr, m) <= Mr, so a sample whose v is not below
        // Mr + w (w = 100 lambda ln 2: the cut of every weight here) cannot carry any -- and against the running minimum of the groups
        // behind it almost nothing passes (a group's best beats the best of g groups with probability 1 / (g + 1)).  Each lane notes the
        // passing samples of ITS row as bits; the wave's OR of them is what gets visited.  The first group of a wave has no running
        // minimum yet: it takes a second pass against its own.
        const double w = -cut / scale;
        double m = INFINITY;
        uint32_t hit_lo = 0, hit_hi = 0;
        auto scan = [&](double thr, bool want_min) __attribute__((always_inline)) {
            // seven samples' pairs at a time: fourteen 16-byte LDS reads requested together (the counter holds fifteen), one wait -- left
            // to itself the compiler waits for every pair (a lone wave pays each of those round trips in full)
#pragma unroll
            for (int q0 = 0; q0 < 32; q0 += 7) {
                d2 sv[7], pv[7];
#pragma unroll
                for (int q = 0; q < 7; ++q)
                    if (q0 + q < 32) { sv[q] = st2[q0 + q]; pv[q] = prow2[q0 + q]; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 7; ++q)
                    if (q0 + q < 32) {     // (+inf for the samples beyond K)
                        const double va = sv[q].x - pv[q].x, vb = sv[q].y - pv[q].y;
                        if (want_min) m = fmin(m, fmin(va, vb));
                        const uint32_t bits = (va < thr ? 1u : 0u) | (vb < thr ? 2u : 0u);
                        if (q0 + q < 16) hit_lo |= bits << (2 * (q0 + q)); else hit_hi |= bits << (2 * (q0 + q) - 32);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        scan(Mr + w, true);
        if (Mr == INFINITY) { hit_lo = 0; hit_hi = 0; scan(m + w, false); }   // (Mr is +inf in every lane, or in none)
        if (lane >= T) { hit_lo = 0; hit_hi = 0; }
        const double Mn = fmin(Mr, m);
        const double so = (Mr == Mn) ? 1.0 : exp((Mn - Mr) * P.inv_lambda);   // (Mr = +inf in front of the first group: exp(-inf) = 0 times D = 0)
        double Dg = 0.0, N0g = 0.0, N1g = 0.0;
        const uint32_t gk0 = P.sample_offset + (uint32_t)(g * 64);
        // the wave's OR of the lanes' bits, through the DPP network (row steps, then the row results across the rows)
        auto wave_or = [](uint32_t v) __attribute__((always_inline)) {
            v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
            v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
            v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);   // row_half_mirror
            v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);   // row_mirror
            return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) | (uint32_t)__builtin_amdgcn_readlane((int)v, 16) |
                   (uint32_t)__builtin_amdgcn_readlane((int)v, 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
        };
        uint64_t want = ((uint64_t)wave_or(hit_hi) << 32) | wave_or(hit_lo);   // (uniform) bit kk: sample kk may have weight in some row
        while (want) {   // (uniform)
            const int kk = __builtin_ctzll(want);
            want &= want - 1;
            const double xs = (Mn - (st[kk] - pf[(size_t)trow * kFusedPitch + kk])) * scale;      // <= 0; -inf beyond K
            const bool cand = xs > cut && lane < T;
            if (!__any(cand)) continue;   // (passed against the old minimum, not against the new one)
            // some row gives sample kk weight: its noise, re-drawn -- lane t draws (sample, step t)
            float f0, f1;
            philox_normal_pair<0>(gk0 + (uint32_t)kk, (uint32_t)trow, tick, P.agent_offset + (uint32_t)a, key0, key1, sigf, f0, f1);
            const double e = cand ? exp2(xs) : 0.0;
            Dg += e;
            N0g = fma(e, (double)f0, N0g);
            N1g = fma(e, (double)f1, N1g);
        }
        const double E0g = (double)es[trow * 2 + 0], E1g = (double)es[trow * 2 + 1];
        const double Cg = (double)min(64, P.K - g * 64);
        // merge into the running tuple (exact: M = min, the old D and N rescaled by exp(-(Mr - Mn) / lambda); E and the count add)
        Dr = Dr * so + Dg; N0r = N0r * so + N0g; N1r = N1r * so + N1g;
        E0r += E0g; E1r += E1g; Cr += Cg; Mr = Mn;
        wave_sync();   // (the next group's prefix stores stay behind this group's reads)
    }
    probe.stop(P);
    // the workgroup's four waves leave ONE tuple per row (256 per row for the merge launch): waves 1-3 hand theirs over through their
    // own LDS regions (nothing else lives there any more), wave 0 merges (exact rescaling) and stores
    double* hand = pf;   // [7][64] of this wave's region
    if (wid != 0) {
        hand[0 * 64 + lane] = Mr; hand[1 * 64 + lane] = Dr; hand[2 * 64 + lane] = N0r; hand[3 * 64 + lane] = N1r;
        hand[4 * 64 + lane] = E0r; hand[5 * 64 + lane] = E1r; hand[6 * 64 + lane] = Cr;
    }
    __syncthreads();   // (every wave gets here exactly once, whatever its number of groups)
    if (wid == 0 && lane < T) {
        double Mo[3], Dn = Dr, N0n = N0r, N1n = N1r, Mn = Mr;
#pragma unroll
        for (int w2 = 1; w2 < 4; ++w2) {
            const double* h2 = reinterpret_cast<const double*>(reinterpret_cast<const char*>(pf) + (size_t)w2 * wave_bytes);
            Mo[w2 - 1] = h2[0 * 64 + lane];
            Mn = fmin(Mn, Mo[w2 - 1]);
        }
        const double s0 = (Mr == Mn) ? 1.0 : exp((Mn - Mr) * P.inv_lambda);   // (a wave without groups: M = +inf, D = N = 0: exp(-inf) = 0)
        Dn *= s0; N0n *= s0; N1n *= s0;
#pragma unroll
        for (int w2 = 1; w2 < 4; ++w2) {
            const double* h2 = reinterpret_cast<const double*>(reinterpret_cast<const char*>(pf) + (size_t)w2 * wave_bytes);
            const double sc = (Mo[w2 - 1] == Mn) ? 1.0 : exp((Mn - Mo[w2 - 1]) * P.inv_lambda);
            Dn += sc * h2[1 * 64 + lane]; N0n += sc * h2[2 * 64 + lane]; N1n += sc * h2[3 * 64 + lane];
            E0r += h2[4 * 64 + lane]; E1r += h2[5 * 64 + lane]; Cr += h2[6 * 64 + lane];
        }
        double* o = part + (((size_t)a * T + lane) * (NB / 4) + blockIdx.x) * kTupleW;
        o[0] = Mn; o[1] = Dn; o[2] = N0n; o[3] = N1n; o[4] = E0r; o[5] = E1r; o[6] = Cr; o[7] = 0.0;
    }
}

hipError_t launch_rollout_fused(const RolloutFusedArgs& a) {
    const dim3 grid(a.NB / 4, a.P.A);
    const unsigned lds = (unsigned)rollout_fused_lds(a.P.T, a.split);
#define MPPI_FUSED_GO(NT, SP) hipLaunchKernelGGL((rollout_fused_kernel<NT, SP>), grid, dim3(SP ? 512 : 256), lds, a.stream, a.P, a.state, a.goal, a.unom, a.tc, a.seed, a.tick, a.tick_ptr, a.part, a.NB)
    if (a.split) { if (a.nterm == 7) MPPI_FUSED_GO(7, true); else MPPI_FUSED_GO(4, true); }
    else { if (a.nterm == 7) MPPI_FUSED_GO(7, false); else MPPI_FUSED_GO(4, false); }
#undef MPPI_FUSED_GO
    return hipGetLastError();
}
#endif  // MPPI_ROLLOUT_FUSED_TU

}  // namespace mppi
