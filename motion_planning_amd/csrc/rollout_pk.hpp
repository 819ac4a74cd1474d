// rollout_pk.hpp -- rollout_pk_kernel: the lane-per-sample rollout of the node's own configuration in MIXED precision,
// TWO samples per lane on the packed-fp32 pipe (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two fp32 operations in the
// issue slot of one fp64 operation).  Runs the tick path of MPPI_STORE_F32 engines when
//     device Philox noise, noise not stored, rk4 + dd_dynamics, Q = diag(q, q, 0) with q > 0, no obstacle grid, T <= 256
// (everything else -- injected noise, stored noise, fp64 storage, the general cost, euler -- stays on rollout_kernel, which
// is fp64 throughout).  Reference: control/src/mppi get_cost2go :127-178, get_cost :180-184, rk4 :39-54, dd_dynamics :23-30.
//
// DEVIATION FORM.  Every per-sample quantity is carried as its difference from the nominal (eps = 0) trajectory, which the
// block's prologue computes in fp64 (nominal_lanes) and leaves in LDS as one 80-byte row per step:
//     dp_i   = hk (clip(un_i + eps_i) - clip(un_i))             wheel-speed deviations (hk = kth dt / 2), exact clip
//     dphi   = dp1 - dp0,  dPs = dp0 + dp1                       rotation / speed deviation of the step
//     th    += 2 dphi                                            heading deviation          [compensated fp32 sum: thf - thc]
//     alpha  = th_start + dphi                                   mid-step heading deviation
//     S = sin alpha, Cm = cos alpha - 1                          series, |alpha| <= 0.5 (guarded per chunk, see below)
//     G  = rho (Pn + dPs) W(phin + dphi),  dG = G - Gn           Simpson-weighted speed and its deviation
//     (a, b) = (G Cm + dG, G S)                                  increment deviation in the nominal mid-step heading frame
//     dX += c1n a - s1n b,  dY += s1n a + c1n b                  position deviation (scaled by sqrt(q/2)) [fp64 running sums]
//     pre += dX (2 Xn + dX) + dY (2 Yn + dY)                     stage cost minus nominal stage cost     [fp64 running sum]
//     ncs += lam un.Sig.eps                                      its noise-cost part                     [fp32 running sum]
// fp32 (packed): the clip, the series, the speed / Simpson-weight deviations, the heading sum (Kahan: round 4 -- it was an fp64
// sum converted each way per step) and the noise cost with its running sum -- 32 packed operations per step for two samples;
// fp64: the position and cost sums, the rotation into the world frame (it feeds the position sum
// directly: as cheap as a packed rotation plus two conversions, and it keeps the increments' rounding out of the sum)
// and the quadratic cost.  Every fp32 rounding error is proportional to a DEVIATION (none to the nominal's magnitude).
// Measured against the fp64 oracle (tools/pk_error_model.py is this arithmetic in numpy; tests replay the kernel):
// |V - V_oracle| <= ~1e-6 typical, <= 7e-6 worst at config 4 -- the class of the fp32 storage of V (5.4e-6 there).
//
// Guard: the series hold for |alpha| <= 0.5.  A chunk of <= 8 steps moves th by at most 8 * 2 hk sqrt2 * 5.53 sigma (the
// Box-Muller radius is bounded), so a chunk whose lanes all start with |th| <= al_guard = 0.5 - that bound runs the
// short series; otherwise (never at the node's sigma: 7 standard deviations) the same step runs series that hold for
// |alpha| <= 2.  The engine selects this kernel only when T times that per-step bound is <= 2 (rollout_pk_applies).
//
// PACK (option "noise_packing", NoisePack in mppi_kernels.hpp): how a Philox call's bits become normals -- 0 three steps per call
// (chunks of six steps), 1 four (16-bit uniforms), 2 hipRAND's own normals, two (both: chunks of eight steps).
// T <= 64: one wave runs the block's nominal prologue; the other three draw their first chunk's noise meanwhile (PACK 0).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "mppi_kernels.hpp"

namespace mppi {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

// (PkRow, the 80-byte per-step row of the nominal trajectory this kernel reads from LDS, is defined in mppi_kernels.hpp: the
// finalize kernel writes the same rows for launches that load their table)

struct RolloutPkArgs {
    DevParams P;
    hipStream_t stream;
    int inline_nominal;   // 1: one wave (T <= 64), 2: four waves (T <= 256); 0: no prologue -- the table is loaded (pkrows, tc)
    uint64_t seed;
    uint32_t tick;
    const uint32_t* tick_ptr;
    const double *state, *goal, *unom;
    double *tc, *base;
    float *dP, *stot, *epart;
    float al_guard;
    const PkRow* pkrows;  // inline_nominal 0: the table the previous tick's finalize kernel left ([A][T])
    int noise_pack;       // option "noise_packing": 0 three steps per Philox call (the default stream), 1 four, 2 hipRAND's normals, two (NoisePack, mppi_kernels.hpp)
    hipEvent_t ev_start, ev_stop;
};
hipError_t launch_rollout_pk(const RolloutPkArgs& a);
// |2 dphi| of one step is at most 2 hk (|e0| + |e1|) <= 2 hk sqrt2 * radius * sigma; the Box-Muller radius of the packing: 5.53 for
// a 22-bit uniform (4.85 for the 16-bit one), 6.66 for hipRAND's 32-bit one
inline double rollout_pk_radius(int noise_pack) { return noise_pack == 2 ? 6.67 : 5.53; }
inline double rollout_pk_step_bound(double kth, double dt, double sigma, int noise_pack) {
    return 2.0 * (0.5 * kth * dt) * 1.41421356237 * rollout_pk_radius(noise_pack) * sigma;
}
// largest |th| at the start of a chunk (<= 8 steps) for which the short series are valid
inline float rollout_pk_guard(double kth, double dt, double sigma, int noise_pack) {
    return (float)(0.5 - 8.0 * rollout_pk_step_bound(kth, dt, sigma, noise_pack));
}
// the long series hold for |alpha| <= 2: the heading deviation can never leave that range over the whole horizon
inline bool rollout_pk_applies(double kth, double dt, double sigma, int T, int noise_pack) {
    return T <= 256 && rollout_pk_guard(kth, dt, sigma, noise_pack) > 0.05f && (T + 1) * rollout_pk_step_bound(kth, dt, sigma, noise_pack) <= 2.0;
}

#ifdef MPPI_ROLLOUT_PK_TU
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 pk_med3(f2 v, float lo, float hi) {
    return f2{__builtin_amdgcn_fmed3f(v.x, lo, hi), __builtin_amdgcn_fmed3f(v.y, lo, hi)};
}

// SPLIT (under-filled launches, T <= 64, the default noise stream): a workgroup of EIGHT waves for the same 512 samples -- waves 0-3 walk
// the dynamics as ever, waves 4-7 draw the noise (Philox + Box-Muller + the per-wave eps sums: 45 % of a chunk's instructions) one chunk
// ahead and hand it over through LDS, a barrier per chunk.  A launch of <= 256 workgroups puts ONE wave of this kernel on a SIMD, and a
// lone wave leaves a third of the SIMD's issue slots empty (4116 cycles per chunk for 2749 cycles of issue; the pair: 3600 -- EXPERIMENTS.md
// 65, 66); the hardware deals a workgroup's waves to the SIMDs in turn (wave i + 4 lands next to wave i: tools/simd_map.hip), so every
// SIMD hosts a drawing wave and the walking wave it feeds.  Same functions, same operands: bit-identical results.
template <int INLINE_NOM, int WAVES, int PACK, bool SPLIT = false>
__global__ __launch_bounds__(SPLIT ? 512 : 256) __attribute__((amdgpu_waves_per_eu(WAVES, 8))) void rollout_pk_kernel(DevParams P, const double* __restrict__ state,
                                                        const double* __restrict__ goal, double* __restrict__ tc,
                                                        float* __restrict__ dP, float* __restrict__ Stot, uint64_t seed,
                                                        uint32_t tick_arg, const uint32_t* __restrict__ tick_ptr,
                                                        float* __restrict__ epart, const double* __restrict__ unom,
                                                        double* __restrict__ base, float al_guard, const PkRow* __restrict__ pkrows) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    PkRow* lt = reinterpret_cast<PkRow*>(smem_raw);  // [T]
    __shared__ double fin_sh[1];                      // the nominal trajectory's final heading (unwrapped)
    static_assert(!SPLIT || (INLINE_NOM != 2 && PACK == 0 && WAVES <= 2), "the split form: T <= 64, the default noise stream, the under-filled instance");
    const int tid = threadIdx.x, a = blockIdx.y, T = P.T;
    const bool producer = SPLIT && __builtin_amdgcn_readfirstlane(tid >> 8) != 0;   // (wave-uniform) SPLIT: waves 4-7 draw, waves 0-3 walk
    ClockProbe probe(P);
    snapshot_inputs(P, state, goal, unom, a);
    const double gx = goal[a * 3 + 0], gy = goal[a * 3 + 1], gth = goal[a * 3 + 2];
    const int lane = tid & 63;
    const int kwave = (int)blockIdx.x * 512 + ((tid >> 6) & 3) * 128;  // this wave's 128 consecutive samples (SPLIT: the pair's)
    const int kA = kwave + 2 * lane;                             // this lane's two: kA, kA + 1
    const bool actA = kA < P.K, actB = kA + 1 < P.K;
    const bool block_full = ((int)blockIdx.x + 1) * 512 <= P.K;  // (uniform)
    const size_t Ks = (size_t)P.Ks, NW = (size_t)P.NWp;
    // the slots of an eps-sum row that are THIS engine's: ceil(K / 64).  (Until round 6 the bound was the row's pitch: a co-scheduled
    // shard whose last wave holds 64 samples wrote the 0 of that wave's second slot one past its own -- harmless in a buffer of
    // its own, but with the shard's sums as columns of the handle's rows that slot is slot 0 of the NEXT row, i.e. the first wave
    // sum of the other shard: the "shared cache line" error of EXPERIMENTS.md 56, root-caused in 57.)
    const size_t nw_own = ((size_t)P.K + 63) >> 6;
    const uint64_t dP_a64 = reinterpret_cast<uint64_t>(dP + (size_t)a * T * Ks);
    float* const dP_a = reinterpret_cast<float*>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(dP_a64 >> 32)) << 32) |
                                                 (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)dP_a64));
    const __amdgpu_buffer_rsrc_t ep_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        epart, 0, (int)min((size_t)0x7FFFFFFF, (size_t)P.A * T * 2 * NW * sizeof(float)), 0x00020000);
    const float hkf = (float)(0.5 * P.kth * P.dt);
    const uint32_t key0 = (uint32_t)seed, key1 = (uint32_t)(seed >> 32);
    const uint32_t ctrA = P.sample_offset + (uint32_t)kA;
    const uint32_t tick = tick_ptr ? *tick_ptr : tick_arg;
    const float sigf = (float)P.sigma;

    constexpr int SPD = NoisePack<PACK>::kSteps;   // steps per Philox draw: 3 (the default stream) | 4 (16-bit packing) | 2 (hipRAND's normals)
    constexpr int U = PACK ? 8 : 6;  // steps per chunk = two (hipRAND's normals: four) Philox draws per sample; wave_sum16 carries the chunk's 12 | 16 eps sums
    float nz[U][4];             // the chunk's noise: [step]{wheel 0 of kA, wheel 0 of kA + 1, wheel 1 of kA, wheel 1 of kA + 1}
    // WAVES <= 2: the instance for UNDER-FILLED launches (at most two waves per SIMD: a rank's share of a sharded controller).  Same
    // arithmetic, operation for operation; the chunk's table rows are requested from LDS at the chunk's top, in front of its Philox
    // draws, and held in registers (120 of them: no room at four waves per SIMD) -- a wave that has the SIMD almost to itself pays
    // every LDS round trip of the per-step reads in full (the fused fp64 kernel gained 18 % from the same move, EXPERIMENTS.md 58).
    constexpr bool PRE = WAVES <= 2;
    PkRow rows[PRE ? U : 1];
    auto load_rows = [&](int t0, int n) __attribute__((always_inline)) {
        if constexpr (PRE) {
#pragma unroll
            for (int j = 0; j < U; ++j)
                if (j < n) rows[j] = lt[t0 + j];
        }
    };
    float tz[SPD][4];
    bool drawn0 = false;   // (wave-uniform) this wave drew its first chunk's noise before the barrier
    double dX[2] = {0.0, 0.0}, dY[2] = {0.0, 0.0}, pre[2] = {0.0, 0.0};
    // the heading deviation: a compensated (Kahan) fp32 sum of the steps' 2 dphi -- thf is what the next step's series take (it was
    // (float) of an fp64 running sum until round 4: a conversion each way and an fp64 add per sample and step), thf - thc the sum to
    // ~2^-46 relative for the terminal cost
    f2 thf = {0.f, 0.f}, thc = {0.f, 0.f};
    // the noise-cost part of the cost prefix, lam (un . Sig) eps summed in fp32 next to the fp64 prefix (~0.007 per step: 2e-8 of
    // rounding over the horizon, three orders below this kernel's V tolerance) -- a conversion and an fp64 add less per sample and step
    f2 ncs = {0.f, 0.f};

    // Box-Muller for the lane's two samples at once (same arithmetic as box_muller(), operation by operation, so the noise
    // is bit-identical to what the update kernel's re-draw and mppi_download_noise produce): the exactly rounded steps run
    // packed and the results land as (kA, kA + 1) pairs -- no register shuffling between the draw and the packed dynamics
    const float nscale = -1.3862943611198906f * (sigf * sigf);
    constexpr float kU1 = PACK ? 1.0f / 65536.0f : 1.0f / 2097152.0f;   // (box_muller | box_muller16)
    auto bm2 = [&](uint32_t a21a, uint32_t manta, uint32_t a21b, uint32_t mantb, f2& w0, f2& w1) __attribute__((always_inline)) {
        const f2 u1 = pk_fma(f2{(float)a21a, (float)a21b}, f2{kU1, kU1}, f2{0.5f * kU1, 0.5f * kU1});
        const f2 lg = f2{__builtin_amdgcn_logf(u1.x), __builtin_amdgcn_logf(u1.y)} * f2{nscale, nscale};
        const f2 r = f2{__builtin_amdgcn_sqrtf(lg.x), __builtin_amdgcn_sqrtf(lg.y)};
        const float ra = __uint_as_float(0x3F800000u | manta), rb = __uint_as_float(0x3F800000u | mantb);
        w0 = f2{__builtin_amdgcn_cosf(ra), __builtin_amdgcn_cosf(rb)} * r;
        w1 = f2{__builtin_amdgcn_sinf(ra), __builtin_amdgcn_sinf(rb)} * r;
    };
    // the noise of steps SPD * d .. SPD * d + SPD - 1 of both samples (philox_normals<PACK> for two counters)
    auto draw3 = [&](uint32_t d, f2 (&w0)[SPD], f2 (&w1)[SPD]) __attribute__((always_inline)) {
        uint32_t oa[4], ob[4];
        philox4x32_10(ctrA, d, tick, P.agent_offset + (uint32_t)a, key0, key1, oa);
        philox4x32_10(ctrA + 1u, d, tick, P.agent_offset + (uint32_t)a, key0, key1, ob);
        if constexpr (PACK == 0) {
            bm2(oa[0] >> 11, (oa[1] >> 9) & 0x7FFFFCu, ob[0] >> 11, (ob[1] >> 9) & 0x7FFFFCu, w0[0], w1[0]);
            bm2(oa[2] >> 11, (oa[3] >> 9) & 0x7FFFFCu, ob[2] >> 11, (ob[3] >> 9) & 0x7FFFFCu, w0[1], w1[1]);
            bm2(((oa[0] & 0x7FFu) << 10) | ((oa[1] & 0x7FFu) >> 1), ((oa[2] & 0x7FFu) << 12) | ((oa[3] & 0x7FEu) << 1),
                ((ob[0] & 0x7FFu) << 10) | ((ob[1] & 0x7FFu) >> 1), ((ob[2] & 0x7FFu) << 12) | ((ob[3] & 0x7FEu) << 1), w0[2], w1[2]);
        } else if constexpr (PACK == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) bm2(oa[i] & 0xFFFFu, (oa[i] >> 9) & 0x7FFF80u, ob[i] & 0xFFFFu, (ob[i] >> 9) & 0x7FFF80u, w0[i], w1[i]);
        } else {   // hipRAND's own transform, sample by sample
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float a0, a1, b0, b1;
                box_muller_hiprand(oa[2 * i], oa[2 * i + 1], sigf, a0, a1);
                box_muller_hiprand(ob[2 * i], ob[2 * i + 1], sigf, b0, b1);
                w0[i] = f2{a0, b0}; w1[i] = f2{a1, b1};
            }
        }
    };
    // (nsteps < U, uniform: the ragged tail only makes the draws its steps need)
    auto draw = [&](int t0, int nsteps) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < U; j += SPD) {
            f2 w0[SPD], w1[SPD];
            if (j < nsteps) draw3((uint32_t)((t0 + j) / SPD), w0, w1);
            else {
#pragma unroll
                for (int i = 0; i < SPD; ++i) { w0[i] = f2{0.f, 0.f}; w1[i] = f2{0.f, 0.f}; }
            }
#pragma unroll
            for (int i = 0; i < SPD; ++i) { nz[j + i][0] = w0[i].x; nz[j + i][1] = w0[i].y; nz[j + i][2] = w1[i].x; nz[j + i][3] = w1[i].y; }
        }
    };
    if constexpr (INLINE_NOM == 0) {
        // the table from memory: the previous tick's finalize kernel computed it for this tick's inputs (nominal_table_lanes)
        const PkRow* src = pkrows + (size_t)a * T;
        typedef float f4 __attribute__((ext_vector_type(4)));
        for (int i = tid; i < T * 5; i += (SPLIT ? 512 : 256)) reinterpret_cast<f4*>(lt)[i] = reinterpret_cast<const f4*>(src)[i];
        if (tid == 0) fin_sh[0] = tc[(size_t)a * T * kTcW + 7];
        if constexpr (PACK == 0 && !SPLIT) {   // the first chunk's noise depends on nothing the table holds: drawn while the loads are in flight
            draw(0, U);
            drawn0 = true;
        }
    } else {
        // the block runs the nominal rollout itself, lanes = timesteps (as rollout_kernel does), and derives the per-step
        // constants of the deviation form from it
        __shared__ double nom_sh[4];
        // (a scalar condition: the two sides are real branches, the noise registers of one are not live through the other)
        const bool prologue_wave = INLINE_NOM == 2 || __builtin_amdgcn_readfirstlane(tid >> 6) == 0;
        if (SPLIT && !prologue_wave) {
            // (the drawing waves start on their first chunk at once, the other walking waves wait at the first chunk's barrier)
        } else
        if (prologue_wave) {
            double row[5], base_t;
            NomExtra ex;
            nominal_lanes<(INLINE_NOM == 2 ? 4 : 1)>(P, state, goal, unom, a, tid, row, base_t, nom_sh, nullptr, &ex);
            if (tid < T) {
                lt[tid] = make_pkrow(P, row, ex, gx, gy);
                if (tid == T - 1) fin_sh[0] = ex.th + ex.h;
                if (blockIdx.x == 0) {  // for mppi_download_value: V = base + Stot - dP
                    base[(size_t)a * T + tid] = base_t;
                    double* o = tc + ((size_t)a * T + tid) * kTcW;
#pragma unroll
                    for (int i = 0; i < 5; ++i) o[i] = row[i];
                }
            }
        } else {
            // T <= 64: one wave runs the nominal rollout, the other three would wait at the barrier behind it -- and the blocks of a CU
            // run in lockstep, so twice per launch every SIMD would host one working wave and three waiting ones (4.1 us of the
            // launch, measured).  They draw the noise of their first chunk meanwhile (it depends on nothing the prologue computes).
            // (the default stream only: with the eight-step chunks of the other packings the 32 noise registers held across the barrier
            // cost spills -- 19 dwords -- and the launch got longer, 92.2 -> 93.8 us)
            if constexpr (PACK == 0) {
                draw(0, U);
                drawn0 = true;
            }
        }
    }
    if constexpr (!SPLIT) __syncthreads();   // (SPLIT: the first chunk's hand-over barrier is this barrier)
    int mk = 0;   // (timeline marks of a diagnostic build; dead code in the product)
    probe.mark(P, mk++);
    // per-wave sums of eps (the E of the softmax floor term, control/src/mppi:193) for the chunk's steps x 2 wheels: the
    // lane's two samples are added first, one 16-value reduce-scatter serves 128 samples.  epart keeps its
    // [A][T][2][Ks/64] layout: the wave's total goes to the slot of its first 64 samples, 0 to the slot of the other 64
    // (the update kernel only ever sums whole chunks of 8192 samples).
    auto eps_sums = [&](int t0, auto full_tag, auto extra_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value, EXTRA = decltype(extra_tag)::value;
        float ev[16];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            ev[2 * j] = FULL ? nz[j][0] + nz[j][1] : (actA ? nz[j][0] : 0.f) + (actB ? nz[j][1] : 0.f);
            ev[2 * j + 1] = FULL ? nz[j][2] + nz[j][3] : (actA ? nz[j][2] : 0.f) + (actB ? nz[j][3] : 0.f);
        }
#pragma unroll
        for (int j = 0; j < (U == 6 ? 2 : 0); ++j) {   // (the four spare slots of a 6-step chunk)
            ev[12 + 2 * j] = !EXTRA ? 0.f : (FULL ? tz[j][0] + tz[j][1] : (actA ? tz[j][0] : 0.f) + (actB ? tz[j][1] : 0.f));
            ev[13 + 2 * j] = !EXTRA ? 0.f : (FULL ? tz[j][2] + tz[j][3] : (actA ? tz[j][2] : 0.f) + (actB ? tz[j][3] : 0.f));
        }
        const float tot = wave_sum16<(EXTRA || U == 8)>(ev, lane);   // (an 8-step chunk fills all sixteen slots)
        const int idx = sum16_index(lane), te = t0 + (idx >> 1), half = lane >> 4;
        const size_t slot = (size_t)(kwave >> 6) + half;
        const bool mine = lane < 32 && idx < (EXTRA ? 16 : 2 * U) && te < T && slot < nw_own;
        const size_t at = (((size_t)a * T + te) * 2 + (idx & 1)) * NW + slot;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(half ? 0.f : tot), ep_rsrc, mine ? (unsigned)(at * 4) : 0xFFFFFFFFu, 0, kDpStoreAux);
    };
    auto step = [&](int t, int j, f2 n0, f2 n1, auto full_tag, bool robust) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
        const PkRow& r = PRE ? rows[PRE ? j : 0] : lt[t];
        {   // dP[t] = the exclusive cost prefix (control/src/mppi:175 as total minus prefix)
            const __amdgpu_buffer_rsrc_t row = __builtin_amdgcn_make_buffer_rsrc(dP_a + (size_t)t * Ks, 0, (int)(Ks * sizeof(float)), 0x00020000);
            const float pa = (float)pre[0] + ncs.x, pb = (float)pre[1] + ncs.y;
            if (FULL) {
                __builtin_amdgcn_raw_buffer_store_b64(u2{__float_as_uint(pa), __float_as_uint(pb)}, row, (unsigned)kA * 4u, 0, kDpStoreAux);
            } else {
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(pa), row, actA ? (unsigned)kA * 4u : 0xFFFFFFFFu, 0, kDpStoreAux);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(pb), row, actB ? (unsigned)kA * 4u + 4u : 0xFFFFFFFFu, 0, kDpStoreAux);
            }
        }
        // EXPLORE + CLIP (control/src/mppi:147-152) as deviations from the clipped nominal
        const f2 dp0 = pk_med3(pk_fma(n0, f2{hkf, hkf}, f2{r.d0, r.d0}), r.lo0, r.hi0);
        const f2 dp1 = pk_med3(pk_fma(n1, f2{hkf, hkf}, f2{r.d1, r.d1}), r.lo1, r.hi1);
        const f2 dphi = dp1 - dp0, dPs = dp0 + dp1;
        const f2 al = thf + dphi;
        {
            const f2 y = pk_fma(f2{2.f, 2.f}, dphi, -thc), t = thf + y;
            thc = (t - thf) - y;
            thf = t;
        }
        f2 S, Cm;
        const f2 z = al * al;
        if (robust) {  // (uniform, practically never) |alpha| <= 2: eight terms each, truncation < 4e-10
            auto c2 = [](float v) { return f2{v, v}; };
            f2 ps = pk_fma(z, c2(-1.f / 1307674368000.f), c2(1.f / 6227020800.f));
            ps = pk_fma(z, ps, c2(-1.f / 39916800.f)); ps = pk_fma(z, ps, c2(1.f / 362880.f)); ps = pk_fma(z, ps, c2(-1.f / 5040.f));
            ps = pk_fma(z, ps, c2(1.f / 120.f)); ps = pk_fma(z, ps, c2(-1.f / 6.f)); ps = pk_fma(z, ps, c2(1.f));
            S = al * ps;
            f2 pc = pk_fma(z, c2(1.f / 20922789888000.f), c2(-1.f / 87178291200.f));
            pc = pk_fma(z, pc, c2(1.f / 479001600.f)); pc = pk_fma(z, pc, c2(-1.f / 3628800.f)); pc = pk_fma(z, pc, c2(1.f / 40320.f));
            pc = pk_fma(z, pc, c2(-1.f / 720.f)); pc = pk_fma(z, pc, c2(1.f / 24.f)); pc = pk_fma(z, pc, c2(-0.5f));
            Cm = z * pc;
        } else {
            S = al * pk_fma(z, pk_fma(z, pk_fma(z, f2{-1.f / 5040.f, -1.f / 5040.f}, f2{1.f / 120.f, 1.f / 120.f}), f2{-1.f / 6.f, -1.f / 6.f}), f2{1.f, 1.f});
            Cm = z * pk_fma(z, pk_fma(z, pk_fma(z, f2{1.f / 40320.f, 1.f / 40320.f}, f2{-1.f / 720.f, -1.f / 720.f}), f2{1.f / 24.f, 1.f / 24.f}), f2{-0.5f, -0.5f});
        }
        // rk4 (control/src/mppi:39-54) for dd_dynamics (:23-30): Simpson bracket (4 + 2 cos phi) times the mid-step heading
        const f2 dW = dphi * pk_fma(f2{r.Cn, r.Cn}, dphi, f2{r.A1, r.A1});
        const f2 Pt = dPs + f2{r.Pn, r.Pn};
        const f2 Aq = Pt * dW;
        const f2 dG = pk_fma(dPs, f2{r.Wn, r.Wn}, Aq);
        const f2 G = pk_fma(Pt, f2{r.Wn, r.Wn}, Aq);
        const f2 av = pk_fma(G, Cm, dG), bv = G * S;
        const double c1n = r.c1n, s1n = r.s1n, X2 = r.X2, Y2 = r.Y2;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const double ad = (double)(i ? av.y : av.x), bd = (double)(i ? bv.y : bv.x);
            dX[i] = fma(c1n, ad, fma(-s1n, bd, dX[i]));
            dY[i] = fma(s1n, ad, fma(c1n, bd, dY[i]));
        }
        // get_cost (control/src/mppi:180-184) minus the nominal stage cost: u = NOMINAL, eps = UNCLIPPED
        ncs = pk_fma(n0, f2{r.w0, r.w0}, pk_fma(n1, f2{r.w1, r.w1}, ncs));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            pre[i] = fma(dX[i], X2 + dX[i], pre[i]);
            pre[i] = fma(dY[i], Y2 + dY[i], pre[i]);
        }
    };
    auto chunk = [&](int t0, int nsteps, auto full_tag) __attribute__((always_inline)) {
        const bool safe = fabsf(thf.x) <= al_guard && fabsf(thf.y) <= al_guard;
#ifdef MPPI_PK_COUNT_ONLY   // `make asm`: the listing tools/valu_mix.py counts holds only what a chunk executes (the long series never run)
        const bool robust = false;
        (void)safe;
#else
        const bool robust = !__all(safe);
#endif
#pragma unroll
        for (int j = 0; j < U; ++j)
            if (j < nsteps) step(t0 + j, j, f2{nz[j][0], nz[j][1]}, f2{nz[j][2], nz[j][3]}, full_tag, robust);
    };
    const int T4 = T - T % U;
    // T = 6 n + 1 or 6 n + 2 (the node's 50): the steps behind the last full chunk ride along with it (their draw is made
    // with the chunk's two, their eps sums fill the reduce-scatter's four spare slots)
    // (a chunk of 8 steps -- PACK 1 -- fills all sixteen slots: its tail is a chunk of its own)
    const bool ride = U == 6 && T4 >= U && (T - T4 == 1 || T - T4 == 2);  // (uniform)
    auto run = [&](auto full_tag) __attribute__((always_inline)) {
        const int t_loop = ride ? T4 - U : T4;
        // (the first chunk stands in front of the loop: the waves that waited at the barrier have drawn its noise there)
        if (t_loop > 0) {
            load_rows(0, U);
            if (!drawn0) draw(0, U);   // (uniform)
            eps_sums(0, full_tag, std::false_type{});
            chunk(0, U, full_tag);
            probe.mark(P, mk++);
        }
        for (int t0 = U; t0 < t_loop; t0 += U) {
            load_rows(t0, U);
            draw(t0, U);
            eps_sums(t0, full_tag, std::false_type{});
            chunk(t0, U, full_tag);
            probe.mark(P, mk++);
        }
        if (ride) {
            const int t0 = T4 - U;
            load_rows(t0, U);
            draw(t0, U);
            {
                f2 w0[SPD], w1[SPD];
                draw3((uint32_t)(T4 / SPD), w0, w1);
#pragma unroll
                for (int i = 0; i < SPD; ++i) {  // steps at or beyond T carry no noise (they are never integrated)
                    const bool in = T4 + i < T;
                    tz[i][0] = in ? w0[i].x : 0.f; tz[i][1] = in ? w0[i].y : 0.f;
                    tz[i][2] = in ? w1[i].x : 0.f; tz[i][3] = in ? w1[i].y : 0.f;
                }
            }
            eps_sums(t0, full_tag, std::true_type{});
            chunk(t0, U, full_tag);
#pragma unroll
            for (int j = 0; j < SPD; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c) nz[j][c] = tz[j][c];
            load_rows(T4, T - T4);
            chunk(T4, T - T4, full_tag);
        } else if (T4 < T) {  // ragged tail: sums of steps at or beyond T are never stored, their noise is never integrated
            load_rows(T4, T - T4);
            draw(T4, T - T4);
            eps_sums(T4, full_tag, std::false_type{});
            chunk(T4, T - T4, full_tag);
        }
    };
    // SPLIT: the same schedule as a list of hand-overs -- (first step, steps drawn, whether the one or two steps behind the last full
    // chunk ride along) -- that both kinds of wave walk, a barrier per entry: a drawing wave draws, sums, leaves the chunk's noise in
    // one of two LDS buffers and arrives; a walking wave arrives, takes the noise into registers and integrates.  The chunk a
    // walking wave integrates is the one its drawing wave finished before the barrier; meanwhile the next one is drawn.
    typedef float f4n __attribute__((ext_vector_type(4)));
    f4n* const nzb = reinterpret_cast<f4n*>(smem_raw + ((size_t)T * sizeof(PkRow) + 15) / 16 * 16) + (size_t)((tid >> 6) & 3) * (U + SPD) * 64 + lane;
    constexpr int kBufStride = 4 * (U + SPD) * 64;   // float4s per buffer (four wave pairs)
    // (the two kinds of wave are separate loops over the same list: neither carries the other's registers)
    auto for_each_hand_over = [&](auto&& f) __attribute__((always_inline)) {
        const int t_loop = ride ? T4 - U : T4;
        if (t_loop > 0) f(0, U, std::false_type{});
        for (int t0 = U; t0 < t_loop; t0 += U) f(t0, U, std::false_type{});
        if (ride) f(T4 - U, U, std::true_type{});
        else if (T4 < T) f(T4, T - T4, std::false_type{});
    };
    auto run_draw = [&](auto full_tag) __attribute__((always_inline)) {
        int ev = 0;
        for_each_hand_over([&](int t0, int n, auto extra_tag) __attribute__((always_inline)) {
            constexpr bool EXTRA = decltype(extra_tag)::value;
            f4n* const buf = nzb + (size_t)(ev & 1) * kBufStride;
            ++ev;
            draw(t0, n);
            if constexpr (EXTRA) {
                f2 w0[SPD], w1[SPD];
                draw3((uint32_t)(T4 / SPD), w0, w1);
#pragma unroll
                for (int i = 0; i < SPD; ++i) {  // steps at or beyond T carry no noise (they are never integrated)
                    const bool in = T4 + i < T;
                    tz[i][0] = in ? w0[i].x : 0.f; tz[i][1] = in ? w0[i].y : 0.f;
                    tz[i][2] = in ? w1[i].x : 0.f; tz[i][3] = in ? w1[i].y : 0.f;
                }
            }
            eps_sums(t0, full_tag, extra_tag);
#pragma unroll
            for (int j = 0; j < U; ++j) buf[j * 64] = f4n{nz[j][0], nz[j][1], nz[j][2], nz[j][3]};
            if constexpr (EXTRA) {
#pragma unroll
                for (int j = 0; j < SPD; ++j) buf[(U + j) * 64] = f4n{tz[j][0], tz[j][1], tz[j][2], tz[j][3]};
            }
            __syncthreads();
        });
    };
    auto run_walk = [&](auto full_tag) __attribute__((always_inline)) {
        int ev = 0;
        for_each_hand_over([&](int t0, int n, auto extra_tag) __attribute__((always_inline)) {
            constexpr bool EXTRA = decltype(extra_tag)::value;
            const f4n* const buf = nzb + (size_t)(ev & 1) * kBufStride;
            ++ev;
            __syncthreads();
            load_rows(t0, n);
#pragma unroll
            for (int j = 0; j < U; ++j) { const f4n v = buf[j * 64]; nz[j][0] = v.x; nz[j][1] = v.y; nz[j][2] = v.z; nz[j][3] = v.w; }
            if constexpr (EXTRA) {
#pragma unroll
                for (int j = 0; j < SPD; ++j) { const f4n v = buf[(U + j) * 64]; tz[j][0] = v.x; tz[j][1] = v.y; tz[j][2] = v.z; tz[j][3] = v.w; }
            }
            chunk(t0, n, full_tag);
            if constexpr (EXTRA) {
#pragma unroll
                for (int j = 0; j < SPD; ++j)
#pragma unroll
                    for (int c = 0; c < 4; ++c) nz[j][c] = tz[j][c];
                load_rows(T4, T - T4);
                chunk(T4, T - T4, full_tag);
            }
            probe.mark(P, mk++);
        });
    };
    if constexpr (SPLIT) {
        if (producer) {
            if (block_full) run_draw(std::true_type{});
            else run_draw(std::false_type{});
            return;   // (behind its last barrier: the walking waves meet no other)
        }
        if (block_full) run_walk(std::true_type{});
        else run_walk(std::false_type{});
    } else {
        if (block_full) run(std::true_type{});
        else run(std::false_type{});
    }
    // terminal cost (control/src/mppi:165-173) minus the nominal's; the theta error is not wrapped beyond rk4's own wrap
    const PkRow& rT = lt[T - 1];
    const double thn = fin_sh[0], inv_f2 = P.lean_inv_f * P.lean_inv_f;
    const double thnw = (thn > M_PI || thn <= -M_PI) ? wrap_theta(thn) : thn;
    const double dthn = thnw - gth;
    float tot[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const double ths = thn + ((double)(i ? thf.y : thf.x) - (double)(i ? thc.y : thc.x));
        const double thw = (ths > M_PI || ths <= -M_PI) ? wrap_theta(ths) : ths;
        const double dths = thw - gth;
        const double term = inv_f2 * (P.p0 * dX[i] * (rT.X2 + dX[i]) + P.p1 * dY[i] * (rT.Y2 + dY[i])) + P.p2 * (dths * dths - dthn * dthn);
        tot[i] = (float)(pre[i] + term + (double)(i ? ncs.y : ncs.x));
    }
    // value_fcn = reverse cumulative sum over t (control/src/mppi:175) = total - exclusive prefix
    if (kDpStoreAux & 16) {   // (sc1 stores, like the rows: nothing of this launch stays dirty in the L2)
        if (block_full) {
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(Stot + (size_t)a * Ks + kA),
                               ((unsigned long long)__float_as_uint(tot[1]) << 32) | __float_as_uint(tot[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (actA) __hip_atomic_store(Stot + (size_t)a * Ks + kA, tot[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (actB) __hip_atomic_store(Stot + (size_t)a * Ks + kA + 1, tot[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else if (block_full) {
        *reinterpret_cast<f2*>(Stot + (size_t)a * Ks + kA) = f2{tot[0], tot[1]};
    } else {
        if (actA) Stot[(size_t)a * Ks + kA] = tot[0];
        if (actB) Stot[(size_t)a * Ks + kA + 1] = tot[1];
    }
    probe.mark(P, mk++);
    probe.stop(P);
}

hipError_t launch_rollout_pk(const RolloutPkArgs& a) {
    dim3 grid((a.P.K + 511) / 512, a.P.A);
    const unsigned lds = (unsigned)((size_t)a.P.T * sizeof(PkRow));
#define MPPI_PK_GO_(IN, W, PK)                                                                                                    \
    do {                                                                                                                      \
        if (a.ev_start)                                                                                                       \
            hipExtLaunchKernelGGL((rollout_pk_kernel<IN, W, PK>), grid, dim3(256), lds, a.stream, a.ev_start, a.ev_stop, 0, a.P, a.state, \
                                  a.goal, a.tc, a.dP, a.stot, a.seed, a.tick, a.tick_ptr, a.epart, a.unom, a.base, a.al_guard, a.pkrows); \
        else                                                                                                                  \
            hipLaunchKernelGGL((rollout_pk_kernel<IN, W, PK>), grid, dim3(256), lds, a.stream, a.P, a.state, a.goal, a.tc, a.dP, a.stot, \
                               a.seed, a.tick, a.tick_ptr, a.epart, a.unom, a.base, a.al_guard, a.pkrows);                    \
    } while (0)
#define MPPI_PK_GO(IN, W) do { if (a.noise_pack == 2) MPPI_PK_GO_(IN, W, 2); else if (a.noise_pack == 1) MPPI_PK_GO_(IN, W, 1); else MPPI_PK_GO_(IN, W, 0); } while (0)
    // (WAVES = 4: the compiler's own allocation, no spills; a fifth wave per SIMD cost spills and measured slower -- EXPERIMENTS.md.
    //  WAVES = 2: the under-filled instance, table rows in registers -- launches of at most two waves per SIMD, the default noise stream)
    const bool under_filled = (long)grid.x * grid.y <= 512 && a.noise_pack == 0;
    // ... and at most ONE wave per SIMD (256 workgroups), T <= 64: the split form -- eight waves per workgroup, four of them drawing the noise
    const bool split = under_filled && (long)grid.x * grid.y <= 256 && a.inline_nominal != 2;
    if (split) {
        const unsigned lds_split = (unsigned)(((size_t)a.P.T * sizeof(PkRow) + 15) / 16 * 16 + (size_t)2 * 4 * (6 + 3) * 64 * 16);
#define MPPI_PK_SPLIT_(IN)                                                                                                       \
        do {                                                                                                                  \
            if (a.ev_start)                                                                                                   \
                hipExtLaunchKernelGGL((rollout_pk_kernel<IN, 2, 0, true>), grid, dim3(512), lds_split, a.stream, a.ev_start, a.ev_stop, 0, a.P, a.state, \
                                      a.goal, a.tc, a.dP, a.stot, a.seed, a.tick, a.tick_ptr, a.epart, a.unom, a.base, a.al_guard, a.pkrows); \
            else                                                                                                              \
                hipLaunchKernelGGL((rollout_pk_kernel<IN, 2, 0, true>), grid, dim3(512), lds_split, a.stream, a.P, a.state, a.goal, a.tc, a.dP, a.stot, \
                                   a.seed, a.tick, a.tick_ptr, a.epart, a.unom, a.base, a.al_guard, a.pkrows);                \
        } while (0)
        if (a.inline_nominal == 1) MPPI_PK_SPLIT_(1); else MPPI_PK_SPLIT_(0);
#undef MPPI_PK_SPLIT_
    } else
    if (under_filled) {
        if (a.inline_nominal == 2) MPPI_PK_GO_(2, 2, 0); else if (a.inline_nominal == 1) MPPI_PK_GO_(1, 2, 0); else MPPI_PK_GO_(0, 2, 0);
    } else
    if (a.inline_nominal == 2) MPPI_PK_GO(2, 4); else if (a.inline_nominal == 1) MPPI_PK_GO(1, 4); else MPPI_PK_GO(0, 4);
#undef MPPI_PK_GO
#undef MPPI_PK_GO_
    return hipGetLastError();
}
#endif  // MPPI_ROLLOUT_PK_TU

}  // namespace mppi
