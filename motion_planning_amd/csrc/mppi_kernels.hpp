// mppi_kernels.hpp -- hand-written gfx950 (CDNA4, wave64) kernels of the MPPI hot path.
//
// Reference being replaced (moribots/motion_planning): control/src/mppi
//   get_cost2go :127-178, get_cost :180-184, rk4 :39-54, dd_dynamics :23-30,
//   update_action :186-208, perform_action :210-213, get_path :85-102.
//
// Pipeline of one control tick (DESIGN.md has the byte accounting):
//   rollout_kernel   1 lane / sample    prologue (T <= 256): the block runs the nominal (eps = 0) rollout,
//                                       lanes = timesteps, into the LDS per-step table; then
//                                       noise -> clip -> RK4 step -> stage cost, all in registers;
//                                       streams the running cost prefix dP[a][t][K] (4 B/step),
//                                       one total per sample and the per-wave sums of eps.  The noise
//                                       itself is written only by the stand-alone mppi_rollout: it is
//                                       a pure function of its Philox counter and is re-drawn on demand.
//   (nominal_kernel  1 block / agent    the same nominal rollout by block scans, only for T > 256)
//   update_kernel    1 block / (chunk,t,a)  per-timestep softmax over K: reads the cost prefix back
//                                       (4 B/step) and eps only for the few samples with weight
//   merge_kernel     1 block / (t,a)    merges chunk partials -> shard partial [A][T][8]
//   finalize_kernel  1 block / agent    merges shard partials (after the RCCL all-gather),
//                                       control update, clip, Savitzky-Golay, clip, plant step, shift
// Small K (the node's own K = 10 ... ~10^4 samples) takes the latency path instead of rollout + update:
//   scan_tick_kernel 1 wave|block / sample   lanes = TIMESTEPS: the trajectory as prefix scans over t,
//                                       cost-to-go as a fourth scan, each timestep's lane folds the
//                                       sample into its running softmax tuple -- V never leaves registers
//
// No MFMA: there is no dense contraction on this path.  All state / cost arithmetic is
// fp64 (lambda = 1e-3 amplifies cost error 1000x inside exp(), fp32 accumulation of V ~ 1e4
// cannot feed it); only the HBM-resident intermediates are stored narrow (fp32) and V is
// stored as an offset from the nominal trajectory's cost-to-go so that fp32 keeps ~1e-6
// absolute accuracy where the softmax weights live:
//     V[t][k] = base[t] + Stot[k] - dP[t][k],   dP[t][k] = sum_{tau<t} (c[tau][k] - c_nom[tau])
// (the reverse cumsum of control/src/mppi:175 written as total minus exclusive prefix, so the
// rollout streams its output step by step and needs no per-lane history).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace mppi {

constexpr int kTupleW = 8;  // {min, D, N0, N1, E0, E1, count, pad}
// Cache-policy bits of what the rollout kernels store (the aux operand of raw_buffer_store on gfx950: 1 = sc0, 2 = nt, 16 = sc1).
// sc1 = write through, no line kept in the XCD's L2: a launch that leaves its cost prefix DIRTY in the write-back L2 pays for the
// write-back when it ends -- 4-5 us of an under-filled launch (25 MB dirty at 125 000 samples: rollout 24.0 -> 19.4 us, tick 39.7 ->
// 34.9; config 3 57.7 -> 53.9), nothing of a launch whose rows leave the L2 on their own (config 4 144.7 / 145.1, config 5 128.2 /
// 128.0); `nt` keeps the rollout's gain and slows the update that reads the rows back (profiles/r5_ab_store_policy.jsonl).
constexpr int kDpStoreAux = 16;
constexpr int kDpStoreAuxF64 = kDpStoreAux;   // ... of the 8-byte rows of fp64 storage (400 MB per tick at config 4: more than the 256 MB Infinity Cache holds)
// The update kernel reads 8-byte rows with non-temporal loads, 4-byte rows with plain ones: the update of fp64 storage reads 400 MB per
// tick at config 4 next to a per-sample total it re-reads for every row; with rows that do not stay in the L2 behind them, that total
// does: update 90.7 -> 80.0 us, tick 213 -> 200 (same box; 4-byte rows: 144.3 / 145.0 on one engine, 135.2 -> 139.1 co-scheduled;
// profiles/r5_ab_f64_memory_policy.jsonl)
constexpr int kTcW = 8;     // {un0, un1, w0, w1, cb, 0, 0, 0}

struct DevParams {
    int A, K, Ks, T;          // Ks: padded row pitch (elements) of eps / dP rows
    int NWp;                  // row pitch (elements) of the per-wave eps sums [A][T][2][NWp]: ceil(Ks / 64) rounded up to a whole number of
                              // 128-byte lines -- every row starts on a line, so a co-scheduled shard's columns of it share no line with its neighbour's
    uint32_t sample_offset;
    uint32_t agent_offset;    // global index of local agent 0: the device-noise streams are keyed by the GLOBAL agent index
    double dt, sigma, lambda, inv_lambda;
    double q0, q1, q2, r0, r1, p0, p1, p2;
    // the off-diagonal terms of the SYMMETRIC parts of Q, R, P1 (mppi_set_weight_matrices: the reference multiplies whole
    // matrices, control/src/mppi:168, :181-184); offdiag = any of them is non-zero (a uniform branch around their arithmetic)
    double q01, q02, q12, r01, p01, p02, p12;
    int offdiag;
    double u_max, kth, rhalf, floor_w;  // kth = wheel_radius / wheel_base, rhalf = wheel_radius / 2
    // the `sig` matrix of get_cost (control/src/mppi:184: lam * u . sig . eps), row-major; sigma above is the
    // std-dev the noise is drawn with (= sig[0,0], :143-146).  sigma * I unless mppi_set_sig_matrix was called.
    double sg00, sg01, sg10, sg11;
    int model;                          // 0: rk4 + dd_dynamics, 1: euler + unicycle_dynamics
    // constants of the rollout kernel's lean (scaled-variable) step, computed once on the host instead of by every lane:
    // lean_f = sqrt(q0 / 2), lean_rho = lean_f * (dt * rhalf / 6) / (kth * dt / 2), lean_inv_f = 1 / lean_f
    double lean_f, lean_rho, lean_inv_f;
    // optional obstacle-grid stage cost (extension, include/mppi_hip.h mppi_set_obstacle_grid)
    const signed char* grid;
    int grid_w, grid_h;
    double grid_res, grid_ox, grid_oy, grid_weight;
    // shader-clock probe (mppi_shader_clock): one lane of the rollout launch's middle block notes how many shader cycles
    // (s_memtime) and constant-rate ticks (s_memrealtime) its wave lived: {cycles, ticks}.  nullptr: off.
    unsigned long long* clk;
    // what the receding-horizon shift puts into the freed last column, [A][2] (control/src/mppi:101: uvec_init[:, 0]; zeros
    // unless mppi_set_shift_fill was called)
    const double* shift_fill;
    // non-null: block 0 of a rollout launch leaves the pre-tick {unom [A][2][T], state [A][3], goal [A][3]} here (what the
    // scan kernel's `prev` is): a co-scheduled tick's V exists only shard by shard, and mppi_download_value re-runs it
    double* snap;
};
// pre-tick snapshot by block 0 of a rollout launch (all threads of the block call it)
__device__ __forceinline__ void snapshot_inputs(const DevParams& P, const double* __restrict__ state, const double* __restrict__ goal,
                                                const double* __restrict__ unom, int a) {
    if (P.snap == nullptr || blockIdx.x != 0) return;
    const int T = P.T;
    for (int i = threadIdx.x; i < 2 * T; i += blockDim.x) P.snap[(size_t)a * 2 * T + i] = unom[(size_t)a * 2 * T + i];
    if (threadIdx.x < 3) {
        P.snap[(size_t)P.A * 2 * T + a * 3 + threadIdx.x] = state[a * 3 + threadIdx.x];
        P.snap[(size_t)P.A * 2 * T + (size_t)P.A * 3 + a * 3 + threadIdx.x] = goal[a * 3 + threadIdx.x];
    }
}
constexpr int kProbeMarks = 30;   // stamps of a diagnostic build's timeline (ClockProbe::mark)
// the finalize kernel's timeline in a diagnostic build: thread 0 of agent 0's workgroup stamps clk[2 + 16 + i]
struct FinProbe {
#ifdef MPPI_PROBE_TIMELINE
    unsigned long long c0; bool on; unsigned long long* out;
    __device__ __forceinline__ FinProbe(const DevParams& P, int a) : c0(clock64()), on(P.clk != nullptr && a == 0 && threadIdx.x == 0), out(P.clk) {}
    __device__ __forceinline__ void mark(int i) { if (on && 16 + i < kProbeMarks) out[2 + 16 + i] = clock64() - c0; }
#else
    __device__ __forceinline__ FinProbe(const DevParams&, int) {}
    __device__ __forceinline__ void mark(int) {}
#endif
};
// the update kernel's: thread 0 of the launch's middle workgroup stamps clk[2 + 24 + i]
struct UpdProbe {
#ifdef MPPI_PROBE_TIMELINE
    unsigned long long c0; bool on; unsigned long long* out;
    __device__ __forceinline__ UpdProbe(const DevParams& P, bool mid) : c0(clock64()), on(P.clk != nullptr && mid && threadIdx.x == 0), out(P.clk) {}
    __device__ __forceinline__ void mark(int i) { if (on && 24 + i < kProbeMarks) out[2 + 24 + i] = clock64() - c0; }
#else
    __device__ __forceinline__ UpdProbe(const DevParams&, bool) {}
    __device__ __forceinline__ void mark(int) {}
#endif
};
struct ClockProbe {
    unsigned long long c0 = 0, w0 = 0;
    bool on;
    __device__ __forceinline__ explicit ClockProbe(const DevParams& P)
        : on(P.clk != nullptr && blockIdx.x == (gridDim.x >> 1) && blockIdx.y == 0 && threadIdx.x == 0) {
        if (on) { c0 = clock64(); w0 = wall_clock64(); }
    }
    __device__ __forceinline__ void stop(const DevParams& P) {
        if (on) { P.clk[0] = clock64() - c0; P.clk[1] = wall_clock64() - w0; }
    }
    // diagnostic builds only (make PROBE=1 -> libmppi_hip_probe.so; tools/probe_timeline.py): shader-cycle stamps of the probe
    // wave at points of its life -- behind the prologue's barrier, behind every chunk -- in clk[2 ...]; the product build
    // compiles this to nothing
    __device__ __forceinline__ void mark(const DevParams& P, int i) {
#ifdef MPPI_PROBE_TIMELINE
        if (on && i < kProbeMarks) P.clk[2 + i] = clock64() - c0;
#else
        (void)P; (void)i;
#endif
    }
};

// weight * cell / 100 of the cell holding (x, y); same arithmetic (divide + floor) as the oracle
__device__ __forceinline__ double obstacle_cost(const DevParams& P, double x, double y) {
    const int ix = (int)floor((x - P.grid_ox) / P.grid_res), iy = (int)floor((y - P.grid_oy) / P.grid_res);
    if (ix < 0 || iy < 0 || ix >= P.grid_w || iy >= P.grid_h) return 0.0;
    return P.grid_weight * ((double)P.grid[ix + iy * P.grid_w] / 100.0);
}

// lam * (u . sig): the row vector that multiplies the (unclipped) noise in get_cost, control/src/mppi:184
__device__ __forceinline__ void cost_noise_weights(const DevParams& P, double un0, double un1, double& w0, double& w1) {
    w0 = P.lambda * (un0 * P.sg00 + un1 * P.sg10);
    w1 = P.lambda * (un0 * P.sg01 + un1 * P.sg11);
}

__device__ __forceinline__ double clampd(double v, double lim) { return fmin(fmax(v, -lim), lim); }
// what the off-diagonal terms add to x' M x: 2 (m01 x y + m02 x z + m12 y z)
__device__ __forceinline__ double cross3(double m01, double m02, double m12, double x, double y, double z) {
    return 2.0 * (m01 * x * y + m02 * x * z + m12 * y * z);
}
// the same clip as ONE fp64 instruction (v_min_f64 with |v|) plus a 32-bit sign copy (v_bfi_b32)
__device__ __forceinline__ double clamp_sym(double v, double lim) { return copysign(fmin(fabs(v), lim), v); }
// 2 * v for a normal, finite v: exponent + 1 (a 32-bit integer add instead of an fp64 instruction)
__device__ __forceinline__ double twice(double v) {
    return __hiloint2double(__double2hiint(v) + 0x00100000, __double2loint(v));
}

// control/src/mppi:52-53 : theta -> (-pi, pi]
__device__ __forceinline__ double wrap_theta(double th) {
    return th - (ceil((th + M_PI) / (2.0 * M_PI)) - 1.0) * 2.0 * M_PI;
}

// sin / cos of a SMALL angle by Taylor series (|phi| <= 0.03 for NTERM 4, <= 0.25 for NTERM 7:
// truncation < 1e-17).  The RK4 stages of the diff-drive model evaluate cos/sin at theta,
// theta + h/2, theta + h with h = dt * kth * (u1 - u0) -- a rotation of the heading vector by
// phi = h/2, so no range reduction and no full-range sincos is needed per step.
template <int NTERM>
__device__ __forceinline__ void small_sincos(double phi, double& s, double& c) {
    const double z = phi * phi;
    if (NTERM == 3) {  // |phi| <= 0.03: truncation <= phi^6/720 < 1.1e-12, phi^7/5040 < 5e-15
        s = phi * fma(z, fma(z, 1.0 / 120.0, -1.0 / 6.0), 1.0);
        c = fma(z, fma(z, 1.0 / 24.0, -0.5), 1.0);
    } else if (NTERM == 4) {
        s = phi * fma(z, fma(z, fma(z, -1.0 / 5040.0, 1.0 / 120.0), -1.0 / 6.0), 1.0);
        c = fma(z, fma(z, fma(z, -1.0 / 720.0, 1.0 / 24.0), -0.5), 1.0);
    } else {
        double ps = fma(z, 1.0 / 6227020800.0, -1.0 / 39916800.0);
        ps = fma(z, ps, 1.0 / 362880.0);
        ps = fma(z, ps, -1.0 / 5040.0);
        ps = fma(z, ps, 1.0 / 120.0);
        ps = fma(z, ps, -1.0 / 6.0);
        s = phi * fma(z, ps, 1.0);
        double pc = fma(z, -1.0 / 87178291200.0, 1.0 / 479001600.0);
        pc = fma(z, pc, -1.0 / 3628800.0);
        pc = fma(z, pc, 1.0 / 40320.0);
        pc = fma(z, pc, -1.0 / 720.0);
        pc = fma(z, pc, 1.0 / 24.0);
        pc = fma(z, pc, -0.5);
        c = fma(z, pc, 1.0);
    }
}

// ---------------------------------------------------------------------------------------------
// block-wide inclusive scan of one double per thread (Hillis-Steele through LDS); NT = blockDim.x
// ---------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ double block_scan_incl(double v, double* sh, double& total) {
    const int tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
#pragma unroll
    for (int off = 1; off < NT; off <<= 1) {
        const double add = (tid >= off) ? sh[tid - off] : 0.0;
        __syncthreads();
        v += add;
        sh[tid] = v;
        __syncthreads();
    }
    total = sh[NT - 1];
    __syncthreads();
    return v;
}

// ---------------------------------------------------------------------------------------------
// nominal_kernel: the eps = 0 rollout of each agent.  It only provides a BASELINE (any per-t
// constant cancels in the per-timestep softmax, control/src/mppi:189), so it is free to use
// parallel prefix sums: for this model RK4 == Simpson's rule on the heading, theta/x/y are
// prefix sums over t and the cost-to-go is a suffix sum.
//   tc[a][t]   = {un0, un1, lam*sig*un0, lam*sig*un1, cb}   cb = -1/2 xQx_nom(t) - [t==T-1] xP1x_nom
//   base[a][t] = nominal cost-to-go  (V = base + dV)
// grid = A blocks x 256 threads, dynamic LDS = T doubles.
// ---------------------------------------------------------------------------------------------
constexpr int kNomThreads = 256;

#ifndef MPPI_ROLLOUT_TU  // non-template kernels are emitted by the engine translation unit only
__global__ __launch_bounds__(kNomThreads) void nominal_kernel(DevParams P, const double* __restrict__ state,
                                                             const double* __restrict__ goal,
                                                             const double* __restrict__ unom,
                                                             double* __restrict__ tc,
                                                             double* __restrict__ base) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* cstage = reinterpret_cast<double*>(smem_raw);  // [T]
    __shared__ double sh[kNomThreads];
    const int a = blockIdx.x, tid = threadIdx.x, T = P.T;
    const double gx = goal[a * 3 + 0], gy = goal[a * 3 + 1], gth = goal[a * 3 + 2];
    double car_x = state[a * 3 + 0], car_y = state[a * 3 + 1], car_th = state[a * 3 + 2];
    for (int b0 = 0; b0 < T; b0 += kNomThreads) {
        const int t = b0 + tid;
        const bool valid = t < T;
        const double un0 = valid ? unom[(a * 2 + 0) * T + t] : 0.0;
        const double un1 = valid ? unom[(a * 2 + 1) * T + t] : 0.0;
        const double u0 = clampd(un0, P.u_max), u1 = clampd(un1, P.u_max);
        const double h = valid ? (P.model == 1 ? P.dt * u1 : P.kth * P.dt * (u1 - u0)) : 0.0;
        double tot_h, tot_x, tot_y;
        const double hin = block_scan_incl<kNomThreads>(h, sh, tot_h);
        const double th = car_th + (hin - h);
        double s0, c0, ix, iy;
        sincos(th, &s0, &c0);
        if (P.model == 1) {  // euler + unicycle
            ix = valid ? P.dt * (c0 * u0) : 0.0;
            iy = valid ? P.dt * (s0 * u0) : 0.0;
        } else {
            double s1, c1, s2, c2;
            sincos(th + 0.5 * h, &s1, &c1);
            sincos(th + h, &s2, &c2);
            const double aa = P.dt * P.rhalf * (u0 + u1) * (1.0 / 6.0);
            ix = valid ? aa * (c0 + 4.0 * c1 + c2) : 0.0;
            iy = valid ? aa * (s0 + 4.0 * s1 + s2) : 0.0;
        }
        const double X = car_x + block_scan_incl<kNomThreads>(ix, sh, tot_x);
        const double Y = car_y + block_scan_incl<kNomThreads>(iy, sh, tot_y);
        if (valid) {
            const double thn = (P.model == 1) ? th + h : wrap_theta(th + h);
            const double dx = X - gx, dy = Y - gy, dth = thn - gth;
            double xqx = P.q0 * dx * dx + P.q1 * dy * dy + P.q2 * dth * dth;
            double uru = P.r0 * un0 * un0 + P.r1 * un1 * un1;
            if (P.offdiag) { xqx += cross3(P.q01, P.q02, P.q12, dx, dy, dth); uru += 2.0 * (P.r01 * un0 * un1); }
            double cst = 0.5 * (xqx + uru);
            if (t == T - 1) {
                cst += P.p0 * dx * dx + P.p1 * dy * dy + P.p2 * dth * dth;
                if (P.offdiag) cst += cross3(P.p01, P.p02, P.p12, dx, dy, dth);
            }
            double* o = tc + ((size_t)a * T + t) * kTcW;
            o[0] = un0; o[1] = un1;
            cost_noise_weights(P, un0, un1, o[2], o[3]);
            o[4] = 0.5 * uru - cst;
            if (t != 0) { o[5] = 0.0; o[6] = 0.0; o[7] = 0.0; }
            else { o[5] = c0; o[6] = s0; o[7] = 0.0; }   // (cos, sin) of the pose's heading: read by every lane of a rollout launch (t = 0: th is the pose's theta exactly)
            cstage[t] = cst;
        }
        car_th += tot_h; car_x += tot_x; car_y += tot_y;
    }
    __syncthreads();
    double carry = 0.0;
    for (int b0 = ((T - 1) / kNomThreads) * kNomThreads; b0 >= 0; b0 -= kNomThreads) {
        const int t = b0 + tid;
        const double v = (t < T) ? cstage[t] : 0.0;
        double tot;
        const double inc = block_scan_incl<kNomThreads>(v, sh, tot);
        if (t < T) base[(size_t)a * T + t] = carry + tot - (inc - v);
        carry += tot;
    }
}
#endif

template <typename R> struct Exp2;
template <> struct Exp2<float> {
    // arguments are <= 0 here: the bare v_exp_f32 (flush-to-zero below 2^-126) is exactly what a
    // softmax weight wants; exp2f() would add denormal-range rescaling around it
    static __device__ __forceinline__ float f(float v) { return __builtin_amdgcn_exp2f(v); }
};
template <> struct Exp2<double> {
    static __device__ __forceinline__ double f(double v) { return exp2(v); }
};

template <typename R>
__device__ __forceinline__ R wave_min(R v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
    return v;
}
template <typename R>
__device__ __forceinline__ R wave_sum(R v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// fp64 over the DPP network instead of the LDS crossbar (ds_bpermute): a double moves as two dwords.
// dpp_mov_f64: lanes without a source (or in a masked-off row) read 0; dpp_mov_f64_keep: they read v itself.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov_f64_keep(double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_last(double v) {  // lane 63's value, through the scalar unit
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
// butterfly inside each 16-lane row (every lane gets its row's result), rows chained by row_bcast:15 / :31,
// lane 63 holds the wave's result and hands it to everyone through the scalar unit
template <> __device__ __forceinline__ double wave_sum<double>(double v) {
    v += dpp_mov_f64<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov_f64<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov_f64<0x141, 0xF>(v);  // row_half_mirror
    v += dpp_mov_f64<0x140, 0xF>(v);  // row_mirror
    v += dpp_mov_f64<0x142, 0xA>(v);  // row_bcast:15 -> rows 1, 3
    v += dpp_mov_f64<0x143, 0xC>(v);  // row_bcast:31 -> rows 2, 3
    return wave_last(v);
}
template <> __device__ __forceinline__ double wave_min<double>(double v) {
    v = fmin(v, dpp_mov_f64_keep<0xB1, 0xF>(v));
    v = fmin(v, dpp_mov_f64_keep<0x4E, 0xF>(v));
    v = fmin(v, dpp_mov_f64_keep<0x141, 0xF>(v));
    v = fmin(v, dpp_mov_f64_keep<0x140, 0xF>(v));
    v = fmin(v, dpp_mov_f64_keep<0x142, 0xA>(v));
    v = fmin(v, dpp_mov_f64_keep<0x143, 0xC>(v));
    return wave_last(v);
}


// Sixteen lanes -- one DPP row -- merge the (at most 16) softmax tuples of ONE row: lane g of the row holds tuple g
// (q = nullptr: none), every lane of the row returns the merged tuple {M, D, N0, N1, E0, E1, count}.  The rescaling
// exp() of the 16 tuples run side by side and the sums are four row-level DPP butterflies each: what a thread
// looping over the tuples does in 16 dependent exp() calls (~10 us for a horizon's rows) takes ~1 us.  Every lane of
// the wave must call it (DPP sources must be active lanes).
__device__ __forceinline__ void merge_row16_vals(bool has, double m, double (&v)[6], double inv_lambda, double (&t)[7]) {
    if (!has) m = INFINITY;
    double M = m;
    M = fmin(M, dpp_mov_f64_keep<0xB1, 0xF>(M));   // quad_perm [1,0,3,2]
    M = fmin(M, dpp_mov_f64_keep<0x4E, 0xF>(M));   // quad_perm [2,3,0,1]
    M = fmin(M, dpp_mov_f64_keep<0x141, 0xF>(M));  // row_half_mirror
    M = fmin(M, dpp_mov_f64_keep<0x140, 0xF>(M));  // row_mirror
    const double sc = !has ? 0.0 : (m == M ? 1.0 : exp((M - m) * inv_lambda));
    if (has) { v[0] *= sc; v[1] *= sc; v[2] *= sc; }
    else { v[0] = v[1] = v[2] = v[3] = v[4] = v[5] = 0.0; }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        v[i] += dpp_mov_f64<0xB1, 0xF>(v[i]);
        v[i] += dpp_mov_f64<0x4E, 0xF>(v[i]);
        v[i] += dpp_mov_f64<0x141, 0xF>(v[i]);
        v[i] += dpp_mov_f64<0x140, 0xF>(v[i]);
    }
    t[0] = M;
#pragma unroll
    for (int i = 0; i < 6; ++i) t[i + 1] = v[i];
}
// q = this lane's tuple or nullptr; safe = any readable tuple: the lanes without one read THAT, so that all seven words of every lane are
// requested at once (behind a guard the count word was one round trip to memory and the rest a second one behind it)
__device__ __forceinline__ void merge_row16_words(bool mine, const double (&w)[7], double inv_lambda, double (&t)[7]) {
    const bool has = mine && w[6] > 0.0;
    double v[6] = {0, 0, 0, 0, 0, 0};
    if (has) { v[0] = w[1]; v[1] = w[2]; v[2] = w[3]; v[3] = w[4]; v[4] = w[5]; v[5] = w[6]; }
    merge_row16_vals(has, has ? w[0] : INFINITY, v, inv_lambda, t);
}
__device__ __forceinline__ void merge_row16(const double* q, const double* safe, double inv_lambda, double (&t)[7]) {
    const double* p = q != nullptr ? q : safe;
    double w[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) w[i] = p[i];
    merge_row16_words(q != nullptr, w, inv_lambda, t);
}
// 64-lane sum with DPP adds only (no LDS traffic): after the four row steps every lane of a
// 16-lane row holds its row sum, row_bcast15 / row_bcast31 chain the rows; LANE 63 holds the total.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum_lane63(float v) {
    v += dpp_mov<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141, 0xF>(v);  // row_half_mirror
    v += dpp_mov<0x140, 0xF>(v);  // row_mirror
    v += dpp_mov<0x142, 0xA>(v);  // row_bcast:15 -> rows 1, 3
    v += dpp_mov<0x143, 0xC>(v);  // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ double wave_sum_lane63(double v) { return wave_sum(v); }
// the fp32 reductions of the update kernel on the same network
template <> __device__ __forceinline__ float wave_sum<float>(float v) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_sum_lane63(v)), 63));
}
template <> __device__ __forceinline__ float wave_min<float>(float v) {
#define MPPI_MIN_STEP(CTRL, MASK) \
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, MASK, 0xF, false)))
    MPPI_MIN_STEP(0xB1, 0xF); MPPI_MIN_STEP(0x4E, 0xF); MPPI_MIN_STEP(0x141, 0xF); MPPI_MIN_STEP(0x140, 0xF);
    MPPI_MIN_STEP(0x142, 0xA); MPPI_MIN_STEP(0x143, 0xC);
#undef MPPI_MIN_STEP
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Sixteen 64-lane sums at once (the eps sums of one 6-step chunk use the first 12 of them) as a
// reduce-scatter: each exchange halves the values a lane still carries.  The partner of a halving must
// hold the same set of values and must be reachable by ONE symmetric DPP pattern:
//   lane^7 (row_half_mirror) and lane^8 (row_ror:8) -- the two big halvings (16 -> 8 -> 4 values).  The bit
//       that tells which half a lane keeps is lane bit 2 resp. bit 3 = the DPP *bank*, so no select is
//       needed at all: one unmasked v_add_f32_dpp forms the pair sum of the "low" value in every lane,
//       a second one with bank_mask 0xA resp. 0xC overwrites it with the pair sum of the "high" value
//       in the lanes that keep that one (2 instructions per kept value instead of 2 selects + 1 add);
//       values 12..15 are padding, their overwrite is skipped (those lanes end with a duplicate);
//   lane^1, lane^2 (quad_perm) -- the two small halvings (4 -> 2 -> 1 values), select + add;
// the two row exchanges that remain are plain adds (v_permlane16_swap, v_permlane32_swap -- gfx950
// additions).  Every lane ends with the total of value sum16_index(lane) (lanes whose index is >= 12:
// a duplicate of value index - 8).  The DPP source operands need two wait states behind the VALU that
// wrote them; the compiler cannot see into the asm, hence the leading s_nop 1 of each block.
__device__ __forceinline__ int sum16_index(int lane) {
    return ((lane & 4) << 1) | ((lane & 8) >> 1) | ((lane & 1) << 1) | ((lane >> 1) & 1);
}
template <bool FULL16 = false>   // FULL16: values 12..15 are real too (four more overwrites in the first halving)
__device__ __forceinline__ float wave_sum16(const float (&v)[16], int lane) {
    const bool b1 = lane & 1, b2 = lane & 2;
    float u[8], w[4], x[2];
    if (FULL16)
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %8, %8 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %9, %9 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %10, %10 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %11, %11 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %4, %12, %12 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %5, %13, %13 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %6, %14, %14 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %7, %15, %15 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %16, %16 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %1, %17, %17 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %2, %18, %18 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %3, %19, %19 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %4, %20, %20 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %5, %21, %21 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %6, %22, %22 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %7, %23, %23 row_half_mirror row_mask:0xf bank_mask:0xa"
        : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]), "=&v"(u[4]), "=&v"(u[5]), "=&v"(u[6]), "=&v"(u[7])
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]),
          "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]));
    else
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %8, %8 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %9, %9 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %10, %10 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %11, %11 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %4, %12, %12 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %5, %13, %13 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %6, %14, %14 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %7, %15, %15 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %16, %16 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %1, %17, %17 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %2, %18, %18 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %3, %19, %19 row_half_mirror row_mask:0xf bank_mask:0xa"
        : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]), "=&v"(u[4]), "=&v"(u[5]), "=&v"(u[6]), "=&v"(u[7])
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]),
          "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]));
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %1, %9, %9 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %2, %10, %10 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %3, %11, %11 row_ror:8 row_mask:0xf bank_mask:0xc"
        : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3])
        : "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]), "v"(u[4]), "v"(u[5]), "v"(u[6]), "v"(u[7]));
#pragma unroll
    for (int i = 0; i < 2; ++i)  // partner lane ^ 1: keep i + 2*b1 (+ ...)
        x[i] = (b1 ? w[i + 2] : w[i]) + dpp_mov<0xB1, 0xF>(b1 ? w[i] : w[i + 2]);
    float y = (b2 ? x[1] : x[0]) + dpp_mov<0x4E, 0xF>(b2 ? x[0] : x[1]);  // partner lane ^ 2: keep b2 (+ ...)
    {
        const unsigned t = __float_as_uint(y);
        const auto p = __builtin_amdgcn_permlane16_swap(t, t, false, false);  // rows 0<->1, 2<->3
        y = __uint_as_float(p[0]) + __uint_as_float(p[1]);
    }
    {
        const unsigned t = __float_as_uint(y);
        const auto p = __builtin_amdgcn_permlane32_swap(t, t, false, false);  // lanes 0-31 <-> 32-63
        y = __uint_as_float(p[0]) + __uint_as_float(p[1]);
    }
    return y;
}

// 64-lane inclusive prefix sum on the DPP network (no LDS crossbar, no barriers): Kogge-Stone inside each
// 16-lane row (row_shr 1, 2, 4, 8), then the row totals chained by row_bcast:15 (rows 1, 3) and
// row_bcast:31 (rows 2, 3).  A double moves as two dwords; lanes without a source add 0.
// one Kogge-Stone step of a 64-lane inclusive scan of unit complex numbers (rotations) under multiplication:
// (c, s) *= the partner's (c, s); lanes without a source lane, or in a masked-off row, multiply by (1, 0)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void dpp_rotate_by_partner(double& c, double& s) {
    const double pc = __hiloint2double(__builtin_amdgcn_update_dpp(0x3FF00000, __double2hiint(c), CTRL, ROW_MASK, 0xF, false),
                                       __builtin_amdgcn_update_dpp(0, __double2loint(c), CTRL, ROW_MASK, 0xF, false));
    const double ps = __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(s), CTRL, ROW_MASK, 0xF, false),
                                       __builtin_amdgcn_update_dpp(0, __double2loint(s), CTRL, ROW_MASK, 0xF, false));
    const double cn = c * pc - s * ps;
    s = s * pc + c * ps;
    c = cn;
}
__device__ __forceinline__ void wave_scan_rotations(double& c, double& s) {
    dpp_rotate_by_partner<0x111, 0xF>(c, s);
    dpp_rotate_by_partner<0x112, 0xF>(c, s);
    dpp_rotate_by_partner<0x114, 0xF>(c, s);
    dpp_rotate_by_partner<0x118, 0xF>(c, s);
    dpp_rotate_by_partner<0x142, 0xA>(c, s);
    dpp_rotate_by_partner<0x143, 0xC>(c, s);
}
__device__ __forceinline__ double wave_scan_incl(double v, int /*lane*/) {
    v += dpp_mov_f64<0x111, 0xF>(v);
    v += dpp_mov_f64<0x112, 0xF>(v);
    v += dpp_mov_f64<0x114, 0xF>(v);
    v += dpp_mov_f64<0x118, 0xF>(v);
    v += dpp_mov_f64<0x142, 0xA>(v);
    v += dpp_mov_f64<0x143, 0xC>(v);
    return v;
}

// inclusive prefix sum over NWAVES * 64 consecutive lanes: shuffles inside each wave, the wave totals
// through LDS (two barriers; none for a single wave).  `total` = sum over all lanes.
// NWAVES == 0: every wave of the block takes part, their number read from blockDim.x (<= 16: sh holds 16 doubles)
template <int NWAVES>
__device__ __forceinline__ double lanes_scan_incl(double v, int t, double* sh, double& total) {
    const int lane = t & 63;
    v = wave_scan_incl(v, lane);
    if (NWAVES == 1) { total = wave_last(v); return v; }
    const int wid = t >> 6;
    if (lane == 63) sh[wid] = v;
    __syncthreads();
    double carry = 0.0, tot = 0.0;
    if (NWAVES == 0) {
        const int nw = (int)blockDim.x >> 6;
        for (int w = 0; w < nw; ++w) { const double x = sh[w]; tot += x; if (w < wid) carry += x; }
    } else {
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) { const double x = sh[w]; tot += x; if (w < wid) carry += x; }
    }
    __syncthreads();
    total = tot;
    return v + carry;
}

// The nominal (eps = 0) rollout with lanes = timesteps: ONE wave for T <= 64 (every scan in registers),
// the block's four waves for T <= 256.  Lane t < T gets its table row {un0, un1, lam*sig*un0,
// lam*sig*un1, cb} and base[t].  All NWAVES * 64 lanes must call it (barriers when NWAVES > 1).
// what the deviation-form rollout (rollout_pk_kernel) needs of the nominal trajectory beyond the table row: lane t's
// clipped wheel speeds, its step's rotation h = 2 phi, the mid-step heading vector and the post-step position
struct NomExtra { double u0c, u1c, h, th, c1, s1, X, Y; };
// nominal_lanes_v: the same on VALUES -- the pose (sx, sy, sth), the goal, and this lane's nominal controls (un0, un1; lanes t >= T
// pass anything) -- for callers whose inputs are not in global memory yet (the finalize kernel preparing the NEXT tick's table).
template <int NWAVES>
__device__ __forceinline__ void nominal_lanes_v(const DevParams& P, double sx, double sy, double sth, double gx, double gy, double gth,
                                                double un0_in, double un1_in, int t, double (&row)[5], double& base_t, double* sh,
                                                double* head0 = nullptr, NomExtra* ex = nullptr) {
    // No contraction the source does not spell out: this function is inlined into kernels of different shapes (every rollout kernel's
    // prologue, the finalize kernel's table for the next tick) and all of them must produce the SAME table bit for bit -- a handle
    // whose engines take the table from different places still equals the one engine exactly (tests: agent split, co-scheduled shards)
#pragma clang fp contract(off)
    const int T = P.T;
    double tot_;
    const bool valid = t < T;
    const double un0 = valid ? un0_in : 0.0;
    const double un1 = valid ? un1_in : 0.0;
    const double u0 = clampd(un0, P.u_max), u1 = clampd(un1, P.u_max);
    const double h = valid ? (P.model == 1 ? P.dt * u1 : P.kth * P.dt * (u1 - u0)) : 0.0;
    const double th = sth + (lanes_scan_incl<NWAVES>(h, t, sh, tot_) - h);
    double s0, c0, ix, iy;
    sincos(th, &s0, &c0);
    if (head0 && t == 0) { head0[0] = c0; head0[1] = s0; }  // lane 0: th = the state's theta exactly (its exclusive scan is 0)
    if (P.model == 1) {  // euler + unicycle: x += dt * cos(theta) * u0
        ix = P.dt * (c0 * u0); iy = P.dt * (s0 * u0);
    } else {
        double s1, c1, s2, c2;
        if (fabs(h) <= 0.5) {  // mid / end headings by a small rotation instead of two more sincos
            double sp, cp;
            small_sincos<7>(0.5 * h, sp, cp);
            c1 = c0 * cp - s0 * sp; s1 = s0 * cp + c0 * sp;
            c2 = c1 * cp - s1 * sp; s2 = s1 * cp + c1 * sp;
        } else {
            sincos(th + 0.5 * h, &s1, &c1);
            sincos(th + h, &s2, &c2);
        }
        const double aa = P.dt * P.rhalf * (u0 + u1) * (1.0 / 6.0);
        ix = aa * (c0 + 4.0 * c1 + c2); iy = aa * (s0 + 4.0 * s1 + s2);
        if (ex) { ex->c1 = c1; ex->s1 = s1; }
    }
    const double X = sx + lanes_scan_incl<NWAVES>(valid ? ix : 0.0, t, sh, tot_);
    const double Y = sy + lanes_scan_incl<NWAVES>(valid ? iy : 0.0, t, sh, tot_);
    if (ex) { ex->u0c = u0; ex->u1c = u1; ex->h = h; ex->th = th; ex->X = X; ex->Y = Y; }
    double cst = 0.0;
    row[0] = un0; row[1] = un1; row[2] = 0.0; row[3] = 0.0; row[4] = 0.0;
    if (valid) {
        const double thn = (P.model == 1) ? th + h : wrap_theta(th + h);
        const double dx = X - gx, dy = Y - gy, dth = thn - gth;
        double xqx = P.q0 * dx * dx + P.q1 * dy * dy + P.q2 * dth * dth;
        double uru = P.r0 * un0 * un0 + P.r1 * un1 * un1;
        if (P.offdiag) { xqx += cross3(P.q01, P.q02, P.q12, dx, dy, dth); uru += 2.0 * (P.r01 * un0 * un1); }
        cst = 0.5 * (xqx + uru);
        if (t == T - 1) {
            cst += P.p0 * dx * dx + P.p1 * dy * dy + P.p2 * dth * dth;
            if (P.offdiag) cst += cross3(P.p01, P.p02, P.p12, dx, dy, dth);
        }
        cost_noise_weights(P, un0, un1, row[2], row[3]);
        row[4] = 0.5 * uru - cst;
    }
    double tot;
    const double inc = lanes_scan_incl<NWAVES>(cst, t, sh, tot);
    base_t = tot - (inc - cst);
}
template <int NWAVES>
__device__ __forceinline__ void nominal_lanes(const DevParams& P, const double* __restrict__ state,
                                              const double* __restrict__ goal, const double* __restrict__ unom,
                                              int a, int t, double (&row)[5], double& base_t, double* sh,
                                              double* head0 = nullptr, NomExtra* ex = nullptr) {
    const int T = P.T;
    const bool valid = t < T;
    nominal_lanes_v<NWAVES>(P, state[a * 3 + 0], state[a * 3 + 1], state[a * 3 + 2], goal[a * 3 + 0], goal[a * 3 + 1], goal[a * 3 + 2],
                            valid ? unom[(a * 2 + 0) * T + t] : 0.0, valid ? unom[(a * 2 + 1) * T + t] : 0.0, t, row, base_t, sh, head0, ex);
}

// One step of the nominal trajectory as the deviation-form rollout (rollout_pk.hpp) reads it from LDS: five 16-byte words.
struct __attribute__((aligned(16))) PkRow {
    float d0, d1, lo0, hi0;   // hk (un_i - clip(un_i)); clip bounds of the deviation: hk (-+u_max - clip(un_i))
    float lo1, hi1, A1, Cn;   // dW = dphi (A1 + Cn dphi): A1 = -2 rho sin phin, Cn = -rho cos phin
    float Wn, Pn, w0, w1;     // rho (4 + 2 cos phin); hk (u0c + u1c); lam (un . Sig) -- the noise-cost weights
    double c1n, s1n;          // nominal mid-step heading
    double X2, Y2;            // 2 * scaled nominal position after the step (relative to the goal)
};
static_assert(sizeof(PkRow) == 80, "PkRow is read as five 16-byte LDS words");
__device__ __forceinline__ PkRow make_pkrow(const DevParams& P, const double (&row)[5], const NomExtra& ex, double gx, double gy) {
#pragma clang fp contract(off)   // (as nominal_lanes_v: the same row whichever kernel derives it)
    const double hk = 0.5 * P.kth * P.dt, phin = 0.5 * ex.h;
    double sp, cp;
    if (fabs(phin) <= 0.25) small_sincos<7>(phin, sp, cp);
    else sincos(phin, &sp, &cp);
    PkRow r;
    r.d0 = (float)(hk * (row[0] - ex.u0c)); r.d1 = (float)(hk * (row[1] - ex.u1c));
    r.lo0 = (float)(hk * (-P.u_max - ex.u0c)); r.hi0 = (float)(hk * (P.u_max - ex.u0c));
    r.lo1 = (float)(hk * (-P.u_max - ex.u1c)); r.hi1 = (float)(hk * (P.u_max - ex.u1c));
    r.A1 = (float)(-2.0 * sp * P.lean_rho); r.Cn = (float)(-cp * P.lean_rho);
    r.Wn = (float)((4.0 + 2.0 * cp) * P.lean_rho); r.Pn = (float)(hk * (ex.u0c + ex.u1c));
    r.w0 = (float)row[2]; r.w1 = (float)row[3];
    r.c1n = ex.c1; r.s1n = ex.s1;
    r.X2 = 2.0 * P.lean_f * (ex.X - gx); r.Y2 = 2.0 * P.lean_f * (ex.Y - gy);
    return r;
}

// The nominal trajectory's table in GLOBAL memory, for rollout launches that load it instead of computing it in every workgroup's
// prologue (INLINE_NOM 0): lane t writes
//     tc[a][t][0..4] = the row,  base[a][t],  pk[a][t] = the deviation-form row (rk4 / diff drive only),
//     tc[a][0][5..6] = (cos, sin) of the pose's heading,  tc[a][0][7] = the trajectory's final heading (unwrapped).
// Written by the finalize kernel for the NEXT tick (from the pose its plant step predicts and the controls it shifted:
// finalize_block) -- valid for a tick whose inputs are those outputs, which the engine tracks on the host (table_valid).
// NWAVES 1: one wave (T <= 64), 0: every thread of the block (T <= 256 <= blockDim.x, barriers inside).
template <int NWAVES>
__device__ __forceinline__ void nominal_table_lanes(const DevParams& P, int a, double sx, double sy, double sth, double gx, double gy,
                                                    double gth, double un0, double un1, int t, double* sh, double* __restrict__ tc,
                                                    double* __restrict__ base, PkRow* __restrict__ pk) {
    const int T = P.T;
    double row[5], base_t;
    NomExtra ex;
    double* row0 = tc + (size_t)a * T * kTcW;
    nominal_lanes_v<NWAVES>(P, sx, sy, sth, gx, gy, gth, un0, un1, t, row, base_t, sh, row0 + 5, &ex);
    if (t < T) {
        double* o = row0 + (size_t)t * kTcW;
#pragma unroll
        for (int i = 0; i < 5; ++i) o[i] = row[i];
        base[(size_t)a * T + t] = base_t;
        if (P.model == 0 && pk != nullptr) pk[(size_t)a * T + t] = make_pkrow(P, row, ex, gx, gy);
        if (t == T - 1) row0[7] = ex.th + ex.h;
    }
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11) -- the generator hipRAND exposes as
// HIPRAND_RNG_PSEUDO_PHILOX4_32_10 -- inlined so that (seed, tick, agent, global sample, t)
// addresses the stream identically on any shard layout and on the CPU twin (oracle/).
// ---------------------------------------------------------------------------------------------
// rounds [R0, R1) of the ten, in place (k0, k1 = the call's key: the round keys are k + r * Weyl); philox4x32_10 is rounds [0, 10)
template <int R0, int R1>
__device__ __forceinline__ void philox_rounds(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = R0; r < R1; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];  // v_mad_u64_u32
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        // three-input xor in one instruction (v_bitop3_b32, truth table 0x96 -- a gfx950 addition)
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32(hi1, c[1], k0 + (uint32_t)r * 0x9E3779B9u, 0x96);
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32(hi0, c[3], k1 + (uint32_t)r * 0xBB67AE85u, 0x96);
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    }
}
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
    philox_rounds<0, 10>(out, k0, k1);
}

// Two N(0, sigma^2) draws from two 21-bit uniforms: Box-Muller in fp32 with u1 = (a + 1/2) / 2^21 in
// (0,1) and the angle u2 = b / 2^21 in [0,1) (both exact in fp32), on the gfx950 transcendental units
// directly -- v_log_f32 (log2), v_sqrt_f32, v_sin_f32 / v_cos_f32 (argument in revolutions, exactly what
// Box-Muller wants).  sigma * sqrt(-2 ln u1) = sqrt(nscale * log2 u1), nscale = -2 ln2 sigma^2.
//   u1: one exact fma, a * 2^-21 + 2^-22;
//   u2: never converted -- the 21 bits are dropped into the mantissa of a float in [1, 2) (`ang_mant` =
//       b << 2) and sin/cos of 1 + u2 revolutions is sin/cos of u2 revolutions;
//   (r cos, r sin): one packed multiply.
// Every operation is exactly rounded or a hardware transcendental of exact inputs, so every kernel
// that calls this with the same words gets bit-identical noise (no contraction-dependent rounding).
typedef float float2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void box_muller(uint32_t a21, uint32_t ang_mant, float nscale, float& e0, float& e1) {
    const float u1 = __builtin_fmaf((float)a21, 1.0f / 2097152.0f, 1.0f / 4194304.0f);
    const float rev = __uint_as_float(0x3F800000u | ang_mant);
    const float r = __builtin_amdgcn_sqrtf(nscale * __builtin_amdgcn_logf(u1));
    float2v cs = {__builtin_amdgcn_cosf(rev), __builtin_amdgcn_sinf(rev)};
    cs *= r;
    e0 = cs.x;
    e1 = cs.y;
}

// The same from ONE word: its low 16 bits the radius uniform u1 = (a + 1/2) / 2^16 (radius <= 4.85 sigma), its high 16
// bits the angle (65 536 directions) -- the 16-bit packing of option "noise_packing" = 1.
__device__ __forceinline__ void box_muller16(uint32_t w, float nscale, float& e0, float& e1) {
    const float u1 = __builtin_fmaf((float)(w & 0xFFFFu), 1.0f / 65536.0f, 1.0f / 131072.0f);
    const float rev = __uint_as_float(0x3F800000u | ((w >> 9) & 0x7FFF80u));
    const float r = __builtin_amdgcn_sqrtf(nscale * __builtin_amdgcn_logf(u1));
    float2v cs = {__builtin_amdgcn_cosf(rev), __builtin_amdgcn_sinf(rev)};
    cs *= r;
    e0 = cs.x;
    e1 = cs.y;
}

// hipRAND's own normals (option "noise_packing" = 2): rocrand_device::detail::box_muller of <rocrand/rocrand_normal.h> restated --
// u = 2^-32 + x 2^-32, v = (2 pi) 2^-32 (1 + y), s = sqrtf(-2 logf(u)), (sin v, cos v) s with the same library functions
// (logf / sqrtf of the device library, __sincosf), then sigma: hiprand_normal4() of a Philox4_32_10 state positioned on the
// call's counter returns {e[0], e[1], e[2], e[3]} / sigma bit for bit (tests/native/hiprand_normal4.hip, compiled by the test).
__device__ __forceinline__ void box_muller_hiprand(uint32_t x, uint32_t y, float sigf, float& e0, float& e1) {
    const float u = 2.3283064e-10f + ((float)x * 2.3283064e-10f);
    const float v = 1.46291807e-09f + ((float)y * 1.46291807e-09f);
    const float s = sqrtf(-2.0f * logf(u));
    float sn, cs;
    __sincosf(v, &sn, &cs);
    e0 = sigf * (sn * s);
    e1 = sigf * (cs * s);
}

// How the 128 bits of one Philox call become normals (option "noise_packing", PACK):
//   0  the default stream: THREE steps per call -- three pairs of 21-bit uniforms (the top 21 bits of the four words, plus the
//      2 x 21 bits assembled from their low 11): 1/3 call per step instead of 1/2;
//   1  FOUR steps per call: word j serves step 4 * draw + j (box_muller16).  A quarter fewer calls and no splicing:
//      the mixed-precision rollout runs 6 % shorter with it (same box, 102.0 -> 96.0 us at 10^6 x 50).  Its price is the
//      distribution's resolution (tails cut at 4.85 sigma instead of 5.53), which is why it is an option and not the default;
//      served where the mixed-precision rollout is (fp32 storage, lane kernels, the node's cost).
//   2  hipRAND's normals themselves: TWO steps per call, hiprand_normal4's four values in order (box_muller_hiprand) -- 32-bit
//      uniforms, radius <= 6.66 sigma, the device library's logf / sqrtf: the literal reading of "noise from hipRAND", at the
//      price of half again as many calls as the default and a longer transform.  Served where 1 is.
template <int PACK> struct NoisePack { static constexpr int kSteps = PACK == 1 ? 4 : (PACK == 2 ? 2 : 3); };
constexpr int kStepsPerDraw = NoisePack<0>::kSteps;
// The noise of global sample `gk`, agent a, steps kSteps * draw .. kSteps * draw + kSteps - 1: e[2j], e[2j+1] = (eps0, eps1)
// of step kSteps * draw + j.
template <int PACK = 0>
__device__ __forceinline__ void philox_normals(uint32_t gk, uint32_t draw, uint32_t tick, uint32_t a, uint32_t key0,
                                               uint32_t key1, float sigf, float (&e)[2 * NoisePack<PACK>::kSteps]) {
    uint32_t o[4];
    philox4x32_10(gk, draw, tick, a, key0, key1, o);
    const float nscale = -1.3862943611198906f * (sigf * sigf);
    if constexpr (PACK == 0) {
        box_muller(o[0] >> 11, (o[1] >> 9) & 0x7FFFFCu, nscale, e[0], e[1]);
        box_muller(o[2] >> 11, (o[3] >> 9) & 0x7FFFFCu, nscale, e[2], e[3]);
        box_muller(((o[0] & 0x7FFu) << 10) | ((o[1] & 0x7FFu) >> 1), ((o[2] & 0x7FFu) << 12) | ((o[3] & 0x7FEu) << 1),
                   nscale, e[4], e[5]);
    } else if constexpr (PACK == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) box_muller16(o[j], nscale, e[2 * j], e[2 * j + 1]);
    } else {
        box_muller_hiprand(o[0], o[1], sigf, e[0], e[1]);
        box_muller_hiprand(o[2], o[3], sigf, e[2], e[3]);
    }
}

// the same stream addressed by (sample, t): only step t's pair of its draw's three (four)
template <int PACK = 0>
__device__ __forceinline__ void philox_normal_pair(uint32_t gk, uint32_t t, uint32_t tick, uint32_t a, uint32_t key0,
                                                   uint32_t key1, float sigf, float& e0, float& e1) {
    constexpr uint32_t kS = NoisePack<PACK>::kSteps;
    uint32_t o[4];
    philox4x32_10(gk, t / kS, tick, a, key0, key1, o);
    const uint32_t j = t % kS;
    if constexpr (PACK == 0) {
        const uint32_t a21 = j == 0 ? o[0] >> 11 : (j == 1 ? o[2] >> 11 : ((o[0] & 0x7FFu) << 10) | ((o[1] & 0x7FFu) >> 1));
        const uint32_t mant = j == 0 ? (o[1] >> 9) & 0x7FFFFCu
                                     : (j == 1 ? (o[3] >> 9) & 0x7FFFFCu : ((o[2] & 0x7FFu) << 12) | ((o[3] & 0x7FEu) << 1));
        box_muller(a21, mant, -1.3862943611198906f * (sigf * sigf), e0, e1);
    } else if constexpr (PACK == 1) {
        box_muller16(j == 0 ? o[0] : (j == 1 ? o[1] : (j == 2 ? o[2] : o[3])), -1.3862943611198906f * (sigf * sigf), e0, e1);
    } else {
        box_muller_hiprand(j == 0 ? o[0] : o[2], j == 0 ? o[1] : o[3], sigf, e0, e1);
    }
}

// ---------------------------------------------------------------------------------------------
// rollout_kernel: MPPI.get_cost2go (control/src/mppi:127-178) for one sample per lane.
//   S      storage type of eps / dP / Stot in HBM (float | double)
//   NTERM  4 | 7: Taylor terms for the per-step heading rotation; 0: full sincos every step
//   PHILOX true: draw eps in-kernel; false: READ the injected eps
//   GENERAL false: the node's cost (Q[2,2] = 0, no obstacle grid) -- every full 4-step chunk is one
//          straight-line basic block (no per-step branch at all: theta is wrapped where it is
//          used, the terminal cost is added after the loop), so the scheduler can interleave the
//          Philox / Box-Muller chains with four steps of fp64 dynamics; true: adds the theta term of
//          the stage cost and the optional obstacle-grid lookup
//   STORE_EPS (PHILOX only) write the drawn eps to HBM.  The tick path does not: the noise is a
//          pure function of (seed, tick, agent, sample, t), so the update kernel regenerates the
//          few values it needs and mppi_download_noise regenerates all of it on demand --
//          8 of the 12 B/step never touch HBM.
// grid = (ceil(K / 256), A), block = 256, LDS = the 40*T-byte per-step table.  Per lane and step: 2 eps + 1 dP element
// through HBM, fully coalesced (consecutive lanes = consecutive k).
// ---------------------------------------------------------------------------------------------
template <typename S, int NTERM, bool PHILOX, bool STORE_EPS, int INLINE_NOM, int MODEL, bool GENERAL>
// (fp64 storage: 4 waves per SIMD -- at 5 the kernel spilled 22 registers around its 8-byte stores; 140.8 -> 136.1 us together with the row-buffer stores)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(sizeof(S) == 8 ? 4 : 5, 8))) void rollout_kernel(DevParams P, const double* __restrict__ state,
                                                     const double* __restrict__ goal,
                                                     double* __restrict__ tc, S* __restrict__ eps,
                                                     S* __restrict__ dP, S* __restrict__ Stot, uint64_t seed,
                                                     uint32_t tick_arg, const uint32_t* __restrict__ tick_ptr,
                                                     int k_first, int k_last, S* __restrict__ epart,
                                                     const double* __restrict__ unom, double* __restrict__ base) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* lt = reinterpret_cast<double*>(smem_raw);  // [T][5] per-step table {un0, un1, w0, w1, cb}
    const int tid = threadIdx.x, a = blockIdx.y, T = P.T;
    ClockProbe probe(P);
    snapshot_inputs(P, state, goal, unom, a);
    // the pose and the goal ONCE, in front of the prologue (they were read again behind its barrier: two more dependent round trips in
    // every wave -- over PCIe when the tick's inputs are fresh and sit in the pinned slot); the prologue wave takes the same values
    const double st_x = state[a * 3 + 0], st_y = state[a * 3 + 1], st_th = state[a * 3 + 2];
    const double g_x = goal[a * 3 + 0], g_y = goal[a * 3 + 1], g_th = goal[a * 3 + 2];
    const uint32_t tick_in = PHILOX ? (tick_ptr ? *tick_ptr : tick_arg) : 0u;
    // LEAN: the node's own cost and model (rk4 diff-drive, Q = diag(q, q, 0) with q > 0, no obstacle grid).
    // Its step is written in scaled variables so that constants fold away (5 fp64 instructions fewer per step):
    //   wheel speeds times half_kd (the table holds half_kd * un, the clip bound is half_kd * u_max):
    //       phi = p1 - p0 directly;
    //   positions relative to the goal and times sqrt(q/2): the stage cost is X^2 + Y^2, nothing else;
    //   fp32 storage only: the heading rotation keeps 3 series terms instead of 4 (truncation < 1.1e-12 per
    //       step, four orders below the rounding of the fp32 prefix it ends up in).
    constexpr bool LEAN = !GENERAL && MODEL == 0 && NTERM != 0;
    constexpr int NT_ROT = (LEAN && NTERM == 4 && sizeof(S) == 4) ? 3 : NTERM;
    // The per-step table lives in LDS: read back as wave-uniform (broadcast) ds_reads that the
    // scheduler can hoist, instead of an s_load + s_waitcnt round trip on every step.
    // (cos, sin) of the state's heading: one sincos per block (the nominal rollout's), not one per lane; [2]: 1 / sqrt(q0 / 2),
    // parked here so that it does not hold a pair of SGPRs through the loop (the kernel sits at the SGPR limit: two more and
    // the compiler carries the dP row descriptors in VGPRs and wraps every store in a waterfall loop)
    __shared__ double head_sh[3];
    if (INLINE_NOM) {
        // The block runs the nominal rollout itself, lanes = timesteps: wave 0 alone for T <= 64
        // (INLINE_NOM 1, ~600 instructions, no barrier), all four waves for T <= 256 (INLINE_NOM 2) -- no
        // separate kernel, no launch boundary in front of the rollout.  The first block also publishes
        // base[] (and tc[]) for mppi_download_value.
        __shared__ double nom_sh[4];
        if (INLINE_NOM == 2 || tid < 64) {
            double row[5], base_t;
            nominal_lanes_v<(INLINE_NOM == 2 ? 4 : 1)>(P, st_x, st_y, st_th, g_x, g_y, g_th, tid < T ? unom[(a * 2 + 0) * T + tid] : 0.0,
                                                      tid < T ? unom[(a * 2 + 1) * T + tid] : 0.0, tid, row, base_t, nom_sh, head_sh, nullptr);
            if (tid == 0) head_sh[2] = P.lean_inv_f;
            if (tid < T) {
#pragma unroll
                for (int i = 0; i < 5; ++i) lt[tid * 5 + i] = (LEAN && i < 2) ? row[i] * (0.5 * P.kth * P.dt) : row[i];
                if (blockIdx.x == 0 && k_first == 0) {
                    base[(size_t)a * T + tid] = base_t;
                    double* o = tc + ((size_t)a * T + tid) * kTcW;
#pragma unroll
                    for (int i = 0; i < 5; ++i) o[i] = row[i];
                }
            }
        }
    } else {
        // the table from memory: nominal_kernel's (T > 256, euler), or the one the previous tick's finalize kernel left for this one
        for (int i = tid; i < T * 5; i += blockDim.x) {
            const double v = tc[((size_t)a * T + i / 5) * kTcW + i % 5];
            lt[i] = (LEAN && i % 5 < 2) ? v * (0.5 * P.kth * P.dt) : v;
        }
        if (tid < 2) head_sh[tid] = tc[(size_t)a * T * kTcW + 5 + tid];
        if (tid == 2) head_sh[2] = P.lean_inv_f;
    }
    __syncthreads();
    int mk = 0;   // (timeline marks of a diagnostic build; dead code in the product)
    probe.mark(P, mk++);
    const int k = k_first + blockIdx.x * 256 + tid;  // this launch covers samples [k_first, k_last)
    const bool active = k < k_last;
    const size_t Ks = (size_t)P.Ks;
    double gx = g_x, gy = g_y;
    const double gth = g_th;
    double x = st_x, y = st_y, th = st_th;
    double c = head_sh[0], s = head_sh[1];
    if (LEAN) {
        const double f = P.lean_f, rho = P.lean_rho;
        x = (x - gx) * f; y = (y - gy) * f;  // position is carried relative to the goal
        gx = 0.0; gy = 0.0;
        c *= rho; s *= rho;  // the Simpson weight of (p0 + p1), in scaled position, rides on the heading vector
    }
    S* eps_a = eps + (size_t)a * T * 2 * Ks + k;
    S* dp = dP + (size_t)a * T * Ks + k;
    // row 0 of this agent's dP block, pinned to the scalar unit: the row descriptors of the loop are derived from it by
    // scalar adds.  (Left to itself the compiler may do this 64-bit product on the VALU -- and then has to wrap every
    // buffer store of the loop in a waterfall loop to get the descriptor back into SGPRs.)
    const uint64_t dP_a64 = reinterpret_cast<uint64_t>(dP + (size_t)a * T * Ks);
    S* const dP_a = reinterpret_cast<S*>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(dP_a64 >> 32)) << 32) |
                                         (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)dP_a64));
    const double half_kd = 0.5 * P.kth * P.dt;             // phi = half_kd * (u1 - u0) = h / 2
    const double sixth_rd = P.dt * P.rhalf * (1.0 / 6.0);  // Simpson weight of (u0 + u1)
    const double hq0 = 0.5 * P.q0, hq1 = 0.5 * P.q1, hq2 = 0.5 * P.q2;
    const double p_max = half_kd * P.u_max;                 // LEAN: clip bound of the scaled wheel speeds
    const size_t NW = (size_t)P.NWp;  // row pitch of the eps sums
    const size_t nw_own = ((size_t)P.K + 63) >> 6;   // ... and the slots of a row that are THIS engine's (a co-scheduled shard's row continues with its neighbour's)
    // raw buffer view of epart (byte-addressed, bounds-checked by the hardware); < 4 GB by construction
    const __amdgpu_buffer_rsrc_t ep_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        epart, 0, (int)min((size_t)0x7FFFFFFF, (size_t)P.A * T * 2 * NW * sizeof(S)), 0x00020000);
    const bool block_full = k_first + (int)(blockIdx.x + 1) * 256 <= k_last;  // uniform

    uint32_t key0 = 0, key1 = 0, ctr0 = 0, tick = 0;
    float sigf = 0.f;
    if (PHILOX) {
        key0 = (uint32_t)seed; key1 = (uint32_t)(seed >> 32);
        ctr0 = P.sample_offset + (uint32_t)k;
        tick = tick_in;
        sigf = (float)P.sigma;
    }

    constexpr int U = 6;  // steps per chunk = two Philox draws (wave_sum16 carries its 12 eps sums)
    S cur[U][2], nxt[U][2];
    auto load_chunk = [&](int t0, S (&buf)[U][2]) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int t = t0 + j;
            const bool ok = active && t < T;
            buf[j][0] = ok ? eps_a[(size_t)(t * 2 + 0) * Ks] : (S)0;
            buf[j][1] = ok ? eps_a[(size_t)(t * 2 + 1) * Ks] : (S)0;
        }
    };
    auto draw_chunk = [&](int t0, S (&buf)[U][2], auto tail_tag) {  // two Philox calls, three steps each
        constexpr bool TAIL = decltype(tail_tag)::value;  // ragged tail: skip the draws that lie wholly beyond T
#pragma unroll
        for (int j = 0; j < U; j += kStepsPerDraw) {
            if (!TAIL || t0 + j < T) {  // (uniform)
                float e[6];
                philox_normals(ctr0, (uint32_t)((t0 + j) / kStepsPerDraw), tick, P.agent_offset + (uint32_t)a, key0, key1, sigf, e);
#pragma unroll
                for (int i = 0; i < kStepsPerDraw; ++i) { buf[j + i][0] = (S)e[2 * i]; buf[j + i][1] = (S)e[2 * i + 1]; }
            } else {
#pragma unroll
                for (int i = 0; i < kStepsPerDraw; ++i) { buf[j + i][0] = (S)0; buf[j + i][1] = (S)0; }
            }
        }
    };
    if (!PHILOX) load_chunk(0, cur);

    double pre = 0.0;  // sum_{tau < t} (c[tau] - c_nom[tau])
    S tl[kStepsPerDraw][2];  // the one or two steps behind the last full chunk when they ride along with it (see `run`)
    auto eps_sums = [&](int t0, auto full_tag, auto extra_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        constexpr bool EXTRA = decltype(extra_tag)::value;  // slots 12..15 carry steps t0 + 6, t0 + 7 (from tl)
        {   // sum_k eps per wave for the chunk's 6 steps x 2 wheels (the E of the softmax floor term,
            // control/src/mppi:193): saves the update kernel from reading eps at all (8 of its 12 B/step)
            // The sums are formed in fp32 in BOTH storage modes: E only enters the control update through the weight floor, as
            // 1e-8 * E / (D + 1e-8 K) (control/src/mppi:193-196), so fp32 rounding of a 64-term sum (~1e-7 relative) moves u by
            // ~1e-15 -- and the fp64 form of this reduction (sixteen serial full-wave fp64 DPP sums per chunk) cost the all-fp64
            // mode a quarter of its rollout launch (round 3: 190.7 us against 107.8 for the same arithmetic under fp32 storage).
            float ev[16];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                ev[2 * j] = (FULL || active) ? (float)cur[j][0] : 0.f;
                ev[2 * j + 1] = (FULL || active) ? (float)cur[j][1] : 0.f;
            }
#pragma unroll
            for (int j = 2 * U; j < 16; ++j) ev[j] = !EXTRA ? 0.f : ((FULL || active) ? (float)tl[(j - 2 * U) >> 1][j & 1] : 0.f);
            const float tot = wave_sum16<EXTRA>(ev, tid & 63);
            const int idx = sum16_index(tid & 63), te = t0 + (idx >> 1);
            const bool mine = (tid & 63) < 16 && idx < (EXTRA ? 16 : 2 * U) && te < T && (size_t)(k >> 6) < nw_own;
            const size_t at = (((size_t)a * T + te) * 2 + (idx & 1)) * NW + (k >> 6);
            if (sizeof(S) == 4) {
                // predication by address instead of by branch: a buffer store whose offset lies beyond
                // num_records is dropped by the hardware, so the chunk stays one basic block and the
                // scheduler may interleave the noise chains with the fp64 dynamics that follow
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(tot), ep_rsrc,
                                                      mine ? (unsigned)(at * 4) : 0xFFFFFFFFu, 0, kDpStoreAux);
            } else if (mine) {
                epart[at] = (S)tot;
            }
        }
    };
    // theta is carried UNWRAPPED inside the loop: only cos/sin of it (carried separately as the heading
    // vector) enter the dynamics, and the reference's per-step wrap to (-pi, pi] (control/src/mppi:52-53)
    // is applied where theta itself is used -- the Q[2,2] stage term and the terminal cost.
    // one step t with the noise (n0, n1): the prefix BEFORE the step goes out, then explore + clip + model + cost
    auto step = [&](int t, S n0, S n1, auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;    // every lane of the block has a sample: plain stores
        {
            {
                const double* tcp = lt + t * 5;
                const double un0 = tcp[0], un1 = tcp[1], w0 = tcp[2], w1 = tcp[3], cb = tcp[4];
                const double e0 = (double)n0, e1 = (double)n1;
                if (FULL || active) {
                    if (PHILOX && STORE_EPS) {
                        eps_a[(size_t)(t * 2 + 0) * Ks] = n0;
                        eps_a[(size_t)(t * 2 + 1) * Ks] = n1;
                    }
                    if (FULL) {
                        // row t of dP as its own buffer: the descriptor is scalar arithmetic (base + t * pitch on
                        // the SALU), the lane offset k * sizeof(S) is loop-invariant -- no per-store 64-bit VALU address
                        // (a row stays below 2 GiB in either storage mode: mppi_create checks)
                        const __amdgpu_buffer_rsrc_t row = __builtin_amdgcn_make_buffer_rsrc(
                            dP_a + (size_t)t * Ks, 0, (int)(Ks * sizeof(S)), 0x00020000);
                        if (sizeof(S) == 4) {
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((float)pre), row, (unsigned)k * 4u, 0, kDpStoreAux);
                        } else {
                            typedef unsigned u2v __attribute__((ext_vector_type(2)));
                            __builtin_amdgcn_raw_buffer_store_b64(u2v{(unsigned)__double2loint(pre), (unsigned)__double2hiint(pre)}, row, (unsigned)k * 8u, 0, kDpStoreAuxF64);
                        }
                    } else {
                        dp[(size_t)t * Ks] = (S)pre;
                    }
                }
                // EXPLORE + CLIP (control/src/mppi:147-152)
                if (LEAN) {
                    // un0, un1 hold half_kd * nominal: p = half_kd * clip(un + eps)
                    const double p0 = clamp_sym(fma(e0, half_kd, un0), p_max), p1 = clamp_sym(fma(e1, half_kd, un1), p_max);
                    const double phi = p1 - p0;
                    double sp, cp;
                    small_sincos<NT_ROT>(phi, sp, cp);
                    // (c, s) is g_scale * (cos, sin)(theta).  Mid-step heading by a rotation; end-of-step heading by
                    // cos(th + 2 phi) = 2 cos(phi) cos(th + phi) - cos(th) (one fma per component; it is applied
                    // once per step behind an exact rotation, so it does not run away like the long recurrence).
                    const double c1 = c * cp - s * sp, s1 = s * cp + c * sp;
                    const double tcp = twice(cp);
                    const double c2 = fma(tcp, c1, -c), s2 = fma(tcp, s1, -s);
                    const double g = (p0 + p1) * (tcp + 4.0);  // Simpson bracket (4 + 2 cos phi), see below
                    x = fma(g, c1, x);
                    y = fma(g, s1, y);
                    th = fma(2.0, phi, th);
                    c = c2; s = s2;
                    double dc = fma(x, x, fma(y, y, cb));
                    dc = fma(w0, e0, dc);
                    dc = fma(w1, e1, dc);
                    pre += dc;
                    return;
                }
                const double u0 = clampd(un0 + e0, P.u_max), u1 = clampd(un1 + e1, P.u_max);
                if (MODEL == 1) {
                    // euler (control/src/mppi:57-58) over unicycle_dynamics (:33-36): x += dt cos(th) u0,
                    // th += dt u1, no wrap; the heading vector is rotated by the step's dt*u1
                    const double phi = P.dt * u1, du = P.dt * u0;
                    x = fma(du, c, x);
                    y = fma(du, s, y);
                    th += phi;
                    if (NTERM == 0) {
                        sincos(th, &s, &c);
                    } else {
                        double sp, cp;
                        small_sincos<NTERM>(phi, sp, cp);
                        const double cn = c * cp - s * sp;
                        s = s * cp + c * sp;
                        c = cn;
                    }
                } else {
                    // rk4 (control/src/mppi:39-54) for dd_dynamics (:23-30): theta_dot is constant
                    // over the step, so the four stages sit at theta, theta+h/2 (twice), theta+h.
                    const double phi = half_kd * (u1 - u0);
                    const double aa = sixth_rd * (u0 + u1);
                    double c1, s1, c2, s2;
                    if (NTERM == 0) {
                        sincos(th + phi, &s1, &c1);
                        sincos(th + 2.0 * phi, &s2, &c2);
                        x = fma(aa, c + 4.0 * c1 + c2, x);
                        y = fma(aa, s + 4.0 * s1 + s2, y);
                    } else {
                        double sp, cp;
                        small_sincos<NTERM>(phi, sp, cp);
                        c1 = c * cp - s * sp; s1 = s * cp + c * sp;
                        c2 = c1 * cp - s1 * sp; s2 = s1 * cp + c1 * sp;
                        // Simpson weights k1 + 2 k2 + 2 k3 + k4: cos(th) + cos(th + 2 phi) = 2 cos(phi) cos(th + phi),
                        // so the bracket is (4 + 2 cos phi) times the mid-step heading
                        const double g = aa * fma(2.0, cp, 4.0);
                        x = fma(g, c1, x);
                        y = fma(g, s1, y);
                    }
                    th += 2.0 * phi;
                    c = c2; s = s2;
                }
                // get_cost (control/src/mppi:180-184) minus the nominal stage cost (cb):
                //   1/2 xQx + 1/2 uRu + lam*sig*(un . eps)  with u = NOMINAL, eps = UNCLIPPED
                const double dx = x - gx, dy = y - gy;
                double dc = fma(hq0 * dx, dx, fma(hq1 * dy, dy, cb));
                dc = fma(w0, e0, dc);
                dc = fma(w1, e1, dc);
                if (GENERAL) {
                    if (hq2 != 0.0 || P.offdiag) {  // Q[2,2] = 0 and Q diagonal in the node
                        const double thw = (MODEL == 0 && (th > M_PI || th <= -M_PI)) ? wrap_theta(th) : th;
                        const double dth = thw - gth;
                        dc = fma(hq2 * dth, dth, dc);
                        if (P.offdiag) dc += 0.5 * cross3(P.q01, P.q02, P.q12, dx, dy, dth);
                    }
                    if (P.grid_weight != 0.0) dc += obstacle_cost(P, x, y);  // extension, off in the node
                }
                pre += dc;
            }
        }
    };
    auto integrate = [&](int t0, auto guard_tag, auto full_tag) __attribute__((always_inline)) {
        constexpr bool GUARD = decltype(guard_tag)::value;  // tail chunk: steps beyond T are skipped
#pragma unroll
        for (int j = 0; j < U; ++j)
            if (!GUARD || t0 + j < T) step(t0 + j, cur[j][0], cur[j][1], full_tag);
    };
    // terminal cost (control/src/mppi:165-173), theta error not wrapped beyond rk4's own wrap; the nominal
    // terminal cost is already inside cb[T-1], and dP[T-1] is the prefix BEFORE the last step, so the
    // sample's terminal cost only enters the total
    auto terminal = [&]() {
        const double thw = (MODEL == 0 && (th > M_PI || th <= -M_PI)) ? wrap_theta(th) : th;
        const double inv_sq = !LEAN ? 1.0 : head_sh[2];  // LEAN: x, y are sqrt(q0 / 2) * position
        const double dx = (x - gx) * inv_sq, dy = (y - gy) * inv_sq, dth = thw - gth;
        pre += P.p0 * dx * dx + P.p1 * dy * dy + P.p2 * dth * dth;
        if (GENERAL && P.offdiag) pre += cross3(P.p01, P.p02, P.p12, dx, dy, dth);
    };
    const int T4 = T - T % U;  // steps covered by full chunks
    // T = 6 n + 1 or 6 n + 2 (the node's T = 50, 20): the one or two steps behind the last full chunk do not get a chunk of their
    // own -- their draw is made with that chunk's two and their eps sums ride in wave_sum16's four spare slots (whose lanes map
    // to steps t0 + 6, t0 + 7 as it is); only the steps themselves remain.
    const bool ride = PHILOX && T4 >= U && (T - T4 == 1 || T - T4 == 2);  // (uniform)
    auto run = [&](auto full_tag) {
        const int t_loop = ride ? T4 - U : T4;
        for (int t0 = 0; t0 < t_loop; t0 += U) {  // full chunks: straight-line code
            if (PHILOX) draw_chunk(t0, cur, std::false_type{});
            else load_chunk(t0 + U, nxt);  // prefetch: HBM latency hides under this chunk's math
            eps_sums(t0, full_tag, std::false_type{});
            integrate(t0, std::false_type{}, full_tag);
            if (!PHILOX) {
#pragma unroll
                for (int j = 0; j < U; ++j) { cur[j][0] = nxt[j][0]; cur[j][1] = nxt[j][1]; }
            }
            probe.mark(P, mk++);
        }
        if (PHILOX && ride) {
            const int t0 = T4 - U;
            draw_chunk(t0, cur, std::false_type{});
            {
                float e[6];
                philox_normals(ctr0, (uint32_t)(T4 / kStepsPerDraw), tick, P.agent_offset + (uint32_t)a, key0, key1, sigf, e);
#pragma unroll
                for (int i = 0; i < kStepsPerDraw; ++i) {  // steps at or beyond T: no noise (they are never integrated)
                    tl[i][0] = T4 + i < T ? (S)e[2 * i] : (S)0;
                    tl[i][1] = T4 + i < T ? (S)e[2 * i + 1] : (S)0;
                }
            }
            eps_sums(t0, full_tag, std::true_type{});
            integrate(t0, std::false_type{}, full_tag);
#pragma unroll
            for (int j = 0; j < U; ++j) {
                cur[j][0] = j < kStepsPerDraw ? tl[j][0] : (S)0;
                cur[j][1] = j < kStepsPerDraw ? tl[j][1] : (S)0;
            }
            integrate(T4, std::true_type{}, full_tag);
        } else if (T4 < T) {  // ragged tail
            if (PHILOX) draw_chunk(T4, cur, std::true_type{});
            eps_sums(T4, full_tag, std::false_type{});
            integrate(T4, std::true_type{}, full_tag);
        }
    };
    if (block_full) run(std::true_type{});   // uniform: all but (at most) the last block of a launch
    else run(std::false_type{});
    terminal();
    // value_fcn = reverse cumulative sum over t (control/src/mppi:175) = total - exclusive prefix
    if (active) {
        if (kDpStoreAux & 16) __hip_atomic_store(Stot + (size_t)a * Ks + k, (S)pre, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (an sc1 store, like the rows)
        else Stot[(size_t)a * Ks + k] = (S)pre;
    }
    probe.mark(P, mk++);
    probe.stop(P);
}

// ---------------------------------------------------------------------------------------------
// update_kernel: the K-reduction of MPPI.update_action (control/src/mppi:187-196).  Per timestep
// the reference needs min_k V, sum_k w and sum_k eps*w with w = exp(-(V - min)/lam) + 1e-8.
//   w = e + floor  =>  sum w = D + floor*K,  sum eps*w = N + floor*E   (floor applied at merge time)
// E = sum eps comes from the rollout kernel's per-wave sums, and e underflows to exactly 0 for
// every sample more than ~0.06 (= 60 lam) above the row minimum, so eps is fetched ONLY for the
// handful of samples that carry weight: the kernel streams 4 B/step (dP) instead of 12.
// Block (t, ch, a) keeps its chunk of v = Stot - dP in registers (kUpdNV 16-byte vectors per lane):
//   pass 1  load, block minimum M;   pass 2  e = exp2((M - v) * log2e/lam), D += e, and for lanes
//   with e > 2^-kCand: N += e * eps  (predicated scalar loads, wave-uniformly skipped otherwise).
// grid = (8 T, ceil(A * chunks of this launch / 8)) x 256 threads; part[a][t][ch] = {M, D, N0, N1, E0, E1, count, 0}.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float uniform_value(float v) { return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v))); }
__device__ __forceinline__ double uniform_value(double v) {
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
// NV 16-byte vectors per lane: chunk = 256 * NV * (16 / sizeof(S)) samples.  8: the streaming shape (54 / 64 VGPRs, eight workgroups per
// CU).  16: chunks twice as long for the sizes where that brings a row down to <= 16 chunk tuples -- the finalize kernel then merges
// them itself and the tick has no merge launch (131 073 ... 262 144 samples in fp32 storage: an N = 4 rank's share of config 4)
constexpr int kUpdNV = 8;
template <typename S, int NV = kUpdNV> struct UpdCfg { static constexpr int VEC = 16 / (int)sizeof(S); static constexpr int CH = 256 * NV * VEC; };

template <typename S, bool REGEN, int PACK = 0, int NV = kUpdNV>
// (fp64 storage: 64 VGPRs, eight blocks per CU -- since the eps sums are read behind the loops and the block minimum sits in scalar
// registers; 70 before, and FORCING 64 then spilled five registers and measured slower on the same box, update 110-117 us against
// 95-103.  Without spills the eighth block changes nothing: 99-100 us for its 400 MB either way.)
__global__ __launch_bounds__(256) void update_kernel(DevParams P, const S* __restrict__ eps,
                                                    const S* __restrict__ dP, const S* __restrict__ Stot,
                                                    double* __restrict__ part, int NCH, int ch_first, int n_local,
                                                    const S* __restrict__ epart, uint64_t seed, uint32_t tick_arg,
                                                    const uint32_t* __restrict__ tick_ptr) {
    using R = S;
    constexpr int VEC = UpdCfg<S, NV>::VEC, CH = UpdCfg<S, NV>::CH;
    typedef S vec_t __attribute__((ext_vector_type(VEC)));
    // XCD-aware block -> (t, chunk) map.  Workgroups go to the 8 XCDs round-robin by linear id, and
    // each XCD has its own L2: id = xcd + 8 * (t + T * group) puts all T blocks that share chunk
    // column (8 * group + xcd) of Stot on ONE XCD, so that chunk crosses into L2 once per tick instead of once
    // per XCD (placement only changes speed, never results).  Chunks are walked from the highest k down:
    // the rollout kernel wrote the high-k columns last, they are what the Infinity Cache still holds.
    // The columns of this launch are the (agent, chunk) pairs, numbered agent-major.
    const int T_ = P.T;
    const int id = blockIdx.x + (int)gridDim.x * blockIdx.y;  // grid = (8 * T, ceil(A * chunks / 8))
    const int xcd = id & 7, q = id >> 3, t = q % T_, grp = q / T_;
    const int col = grp * 8 + xcd;
    UpdProbe uprobe(P, id == (int)(gridDim.x * gridDim.y) / 2);   // (diagnostic builds only)
    if (col >= P.A * n_local) return;
    const int a = col / n_local, local = col % n_local;
    const int ch = ch_first + n_local - 1 - local;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const size_t Ks = (size_t)P.Ks;
    const S* v_row = dP + ((size_t)a * P.T + t) * Ks;   // exclusive cost prefix of row t
    const S* s_row = Stot + (size_t)a * Ks;               // per-sample totals (L2-resident, re-read per t)
    const S* e0_row = eps + (((size_t)a * P.T + t) * 2 + 0) * Ks;
    const S* e1_row = e0_row + Ks;
    const int k_begin = ch * CH;
    const int k_end = min(P.K, k_begin + CH);
    double* o = part + (((size_t)a * P.T + t) * NCH + ch) * kTupleW;
    if (k_end <= k_begin) {  // empty chunk (uniform)
        if (tid == 0) { o[0] = INFINITY; o[1] = 0; o[2] = 0; o[3] = 0; o[4] = 0; o[5] = 0; o[6] = 0; o[7] = 0; }
        return;
    }
    __shared__ R red[4][6];

    // pass 1: the chunk into registers, lane minimum
    S v[NV][VEC];
    R m = (R)INFINITY;
    constexpr bool kNtRows = sizeof(S) == 8;
    if (k_end - k_begin == CH) {
        // (uniform) a whole chunk -- every chunk of a row but its last: no per-lane bounds, so ALL of the chunk's row loads are in flight
        // before the first is waited for.  (Round 5: behind the per-lane guard of the general path below each vector's two loads sit in
        // their own basic block with their own s_waitcnt vmcnt(0) -- eight serial round trips per workgroup, one 16-byte load per lane
        // in flight: that, not the DRAM, was what held this kernel at 5.4 TB/s in 4-byte rows and 4.8 in 8-byte ones.)
        // In flight per lane: the HBM loads of up to eight vectors of the row (32 registers), and the per-sample totals (L2 hits) in
        // groups of kSvGroup behind them -- sized so that the kernel keeps its eight workgroups per CU.
        constexpr int kPvDepth = NV < 8 ? NV : 8;
        constexpr int kSvGroup = 4 < kPvDepth ? 4 : kPvDepth;
#pragma unroll
        for (int h = 0; h < NV; h += kPvDepth) {
            vec_t pv[kPvDepth];
#pragma unroll
            for (int j = 0; j < kPvDepth; ++j) {
                const int k = k_begin + ((h + j) * 256 + tid) * VEC;
                pv[j] = kNtRows ? __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(v_row + k)) : *reinterpret_cast<const vec_t*>(v_row + k);
            }
#pragma unroll
            for (int g = 0; g < kPvDepth; g += kSvGroup) {
                vec_t sv[kSvGroup];
#pragma unroll
                for (int j = 0; j < kSvGroup; ++j) sv[j] = *reinterpret_cast<const vec_t*>(s_row + k_begin + ((h + g + j) * 256 + tid) * VEC);
                if (kSvGroup < kPvDepth) __builtin_amdgcn_sched_barrier(0);   // (the next group's loads stay behind this group's arithmetic)
#pragma unroll
                for (int j = 0; j < kSvGroup; ++j) {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) v[h + g + j][i] = sv[j][i] - pv[g + j][i];
#pragma unroll
                    for (int i = 0; i < VEC; ++i) m = fmin(m, v[h + g + j][i]);
                }
                if (kSvGroup < kPvDepth) __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int k = k_begin + (j * 256 + tid) * VEC;
        if (k + VEC <= k_end) {  // rows are 256-byte aligned and k % VEC == 0: 16-byte aligned loads
            const vec_t pv = kNtRows ? __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(v_row + k)) : *reinterpret_cast<const vec_t*>(v_row + k);
            const vec_t sv = *reinterpret_cast<const vec_t*>(s_row + k);
#pragma unroll
            for (int i = 0; i < VEC; ++i) v[j][i] = sv[i] - pv[i];
        } else {
#pragma unroll
            for (int i = 0; i < VEC; ++i) v[j][i] = (k + i < k_end) ? s_row[k + i] - v_row[k + i] : (S)INFINITY;
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) m = fmin(m, v[j][i]);
    }
    uprobe.mark(0);   // the chunk is in registers (every load waited for)
    m = wave_min(m);
    if (lane == 0) red[wid][0] = m;
    __syncthreads();
    uprobe.mark(1);
    // (the block minimum is uniform: kept in scalar registers -- in the fp64 mode the two vector registers it would hold are what
    // stands between seven and eight blocks per CU)
    const R M = uniform_value(fmin(fmin(red[0][0], red[1][0]), fmin(red[2][0], red[3][0])));
    __syncthreads();

    // pass 2: weights relative to the block minimum; eps only where the weight is representable
    const R scale = (R)(P.inv_lambda * 1.4426950408889634);  // log2(e) / lambda
    // log2 of the smallest weight whose noise is fetched.  D sums every weight of every wave-vector (64 lanes x one 16-byte
    // vector) that holds at least one weight above the cut (vectors without one are skipped whole, see the loop); a weight
    // below the cut only misses its e * eps in N.  fp32 mode: 2^-40 -- the sums are fp32 and the block minimum carries e = 1, so a dropped term is below
    // 1e-12 of D, five orders under the sum's own rounding unit (at 2^-32 two shardings of the same samples still differed by
    // a few 1e-10 in u -- the cut is relative to the CHUNK's minimum -- which the split-invariance tests see; round 2 cut at
    // 2^-80 and fetched twice as many); fp64 mode keeps everything down to 2^-100.
    const R cand = (R)(sizeof(S) == 4 ? -40.0 : -100.0);
    R D = 0;
    double Na0 = 0.0, Na1 = 0.0;
    // fp32 storage: the weights of the CANDIDATES -- the only samples that carry weight at all -- are formed in fp64,
    //     e = exp2(((double)M - (double)v) log2e / lambda)        (M - v is exact in fp64: two fp32 values)
    // and summed in fp64 (Dc; their noise products Na0 / Na1 were fp64 sums already): v_exp_f32 gives every weight its own ~1e-7 of
    // relative error, and M is the CHUNK's minimum -- another way of cutting the samples into chunks, shards or ranks moves M, and
    // with it every weight's error: controls of N = 1 and N = 8 agreed to 1e-8, not to the 1e-10 SURVEY 8d-4 asks for.  In fp64 the
    // weight is the same function of (M - v) to 1e-16 whatever M is, and the tuple merge's rescaling is exact: shard-count-invariant
    // to rounding.  The fp32 sum D keeps the non-candidates of the vectors that were not skipped (< 2^-40 of the chunk's best each).
    // Far from the goal a row has a handful of candidates; parked at it a few per cent: one fp64 exp2 next to each one's Philox re-draw.
    constexpr bool kWideCand = sizeof(S) == 4;
    const double scale64 = P.inv_lambda * 1.4426950408889634, M64 = (double)M;
    double Dc = 0.0;
    // REGEN (eps was never stored): a sample that carries weight needs its noise re-drawn -- one Philox call.  Far from the
    // goal a row has a handful of such samples; parked AT the goal a few per cent of a row carry weight.  The loop only
    // NOTES them -- (index, weight) appended to a queue in LDS -- and the re-draws happen afterwards, spread evenly over the
    // lanes: ceil(candidates / 64) Philox rounds per wave.  (Round 2 kept eight slots per lane: the rounds were the busiest
    // lane's count, ~6 where the average lane had 2.  Round 3 kept ONE queue per block behind an LDS counter: one atomic with
    // return per wave and 16-byte vector -- eight dependent LDS round trips per wave in the parked regime, where every vector
    // holds candidates.  Round 4: one queue per WAVE, its fill count a wave-uniform register: no atomic, no barrier; the
    // block's vectors are dealt to the waves in 1-KB pieces, so the four queues fill evenly.)  A wave whose queue overflows
    // (sigma = 0, a flat cost: every sample carries weight) walks its own values again from L2 and re-draws each candidate
    // in place.
    constexpr int kQueue = (sizeof(S) == 8 ? 1024 : 2048) * NV / kUpdNV;   // (per sample of the chunk as before)
    constexpr int kQW = kQueue / 4;   // per wave
    __shared__ uint32_t q_k[REGEN ? kQueue : 1];
    __shared__ R q_e[REGEN ? kQueue : 1];
    int n_w = 0;   // (wave-uniform) candidates this wave has noted
    const uint32_t tick_now = REGEN ? (tick_ptr ? *tick_ptr : tick_arg) : 0u;
    // a candidate's weight from what the queue holds for it: its value v (fp32 storage) or the weight itself (fp64 storage)
    auto cand_weight = [&](R q) -> double {
        if constexpr (kWideCand) { const double e = exp2((M64 - (double)q) * scale64); Dc += e; return e; }
        else return (double)q;
    };
    auto redraw = [&](uint32_t kk, R q) {
        float f0, f1;  // the same Philox counter the rollout used for this (sample, step)
        philox_normal_pair<PACK>(P.sample_offset + kk, (uint32_t)t, tick_now, P.agent_offset + (uint32_t)a, (uint32_t)seed, (uint32_t)(seed >> 32),
                           (float)P.sigma, f0, f1);
        const double e = cand_weight(q);
        Na0 = fma(e, (double)(S)f0, Na0);   // exact products (fp64 storage) / fp64 products, fp64 sums: independent of how the queue orders them
        Na1 = fma(e, (double)(S)f1, Na1);
    };
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int k = k_begin + (j * 256 + tid) * VEC;
        R xs[VEC], es[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) xs[i] = (M - v[j][i]) * scale;  // <= 0; -inf for the padding
        // A wave's 64 vectors (256 samples) none of which reaches the cut are skipped whole: no exp, nothing added to D -- at most
        // 256 * 2^-40 of the chunk's best weight per skip, 2^-27 of D over a chunk in the worst case (an eighth of D's fp32
        // rounding unit; 2^-92 in fp64 mode).  Far from the goal that is almost every vector: the kernel is HBM-bound either
        // way, but the instructions it does not issue are free slots for a rollout co-scheduled next to it.
        unsigned long long bal[VEC];
        int n_here = 0;
#pragma unroll
        for (int i = 0; i < VEC; ++i) { bal[i] = __ballot(xs[i] > cand); n_here += (int)__popcll(bal[i]); }
        if (n_here == 0) continue;   // (uniform)
#pragma unroll
        for (int i = 0; i < VEC; ++i) { es[i] = Exp2<R>::f(xs[i]); D += (kWideCand && xs[i] > cand) ? (R)0 : es[i]; }
        if (REGEN) {
            if (n_here) {  // (uniform) skipped for almost every vector while the robot is far from its goal
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const int pos = n_w + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal[i] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal[i], 0u));
                    if (xs[i] > cand && pos < kQW) { q_k[wid * kQW + pos] = (uint32_t)(k + i); q_e[wid * kQW + pos] = kWideCand ? (R)v[j][i] : es[i]; }
                    n_w += (int)__popcll(bal[i]);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < VEC; ++i)
                if (xs[i] > cand) {
                    const double e = cand_weight(kWideCand ? (R)v[j][i] : es[i]);
                    Na0 = fma(e, (double)e0_row[k + i], Na0);
                    Na1 = fma(e, (double)e1_row[k + i], Na1);
                }
        }
    }
    uprobe.mark(2);   // weights formed
    if (REGEN) {
        // the wave reads back what its own lanes queued: LDS operations of one wave complete in order, the fence keeps the compiler from
        // moving the reads up
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (n_w <= kQW) {   // (wave-uniform)
            for (int q = lane; q < n_w; q += 64) redraw(q_k[wid * kQW + q], q_e[wid * kQW + q]);
        } else {  // (rare) every candidate of the lane's own values, in place
#pragma unroll 1
            for (int idx = 0; idx < NV * VEC; ++idx) {
                const int k = k_begin + ((idx / VEC) * 256 + tid) * VEC + idx % VEC;
                if (k < k_end) {
                    const R vk = s_row[k] - v_row[k];
                    const R x = (M - vk) * scale;
                    if (x > cand) redraw((uint32_t)k, kWideCand ? vk : Exp2<R>::f(x));
                }
            }
        }
    }
    // sum_k e * eps across the block in fp64: a thread holds a few products at most (the queue spreads the candidates over
    // the block -- a handful of them all sit in wave 0), and an fp32 tree would let the row's dominant term absorb the small
    // ones differently for every way of splitting the samples over chunks / shards
    // E = sum_k eps of this chunk from the per-wave sums (CH/64 entries per wheel, a few hundred bytes) -- formed here, behind the loops,
    // so that it does not hold registers through them
    R E0 = 0, E1 = 0;
    {
        const size_t NW = (size_t)P.NWp;
        const S* ep = epart + (((size_t)a * P.T + t) * 2) * NW;
        const int w_begin = k_begin >> 6, w_end = (k_end + 63) >> 6;
        for (int w = w_begin + tid; w < w_end; w += 256) { E0 += ep[w]; E1 += ep[NW + w]; }
    }
    uprobe.mark(3);   // re-draws done, eps sums read
    __shared__ double redN[4][3];
    const double N0d = wave_sum(Na0), N1d = wave_sum(Na1);
    const double Dd = wave_sum((double)D + Dc);     // (the lane's fp32 sum of non-candidates + its candidates' fp64 weights)
    E0 = wave_sum(E0); E1 = wave_sum(E1);
    if (lane == 0) { redN[wid][2] = Dd; redN[wid][0] = N0d; redN[wid][1] = N1d; red[wid][4] = E0; red[wid][5] = E1; }
    __syncthreads();
    if (tid < 5) {
        const int c = tid + 1;
        const int cn = c == 1 ? 2 : c - 2;
        o[c] = (c <= 3) ? redN[0][cn] + redN[1][cn] + redN[2][cn] + redN[3][cn]
                        : (double)red[0][c] + (double)red[1][c] + (double)red[2][c] + (double)red[3][c];
    } else if (tid == 5) {
        o[0] = (double)M; o[6] = (double)(k_end - k_begin); o[7] = 0.0;
    }
    uprobe.mark(4);
}

// per-wave sums of eps for noise that did not come out of a rollout (mppi_upload_noise followed
// directly by mppi_update): same layout as the rollout kernel's epart.  grid = (ceil(K/256), T*2, A)
template <typename S>
__global__ __launch_bounds__(256) void eps_wavesum_kernel(DevParams P, const S* __restrict__ eps, S* __restrict__ epart) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    const size_t row = (size_t)blockIdx.z * P.T * 2 + blockIdx.y, Ks = (size_t)P.Ks, NW = (size_t)P.NWp;
    const S val = (k < P.K) ? eps[row * Ks + k] : (S)0;
    const S sum = wave_sum_lane63(val);
    if ((threadIdx.x & 63) == 63 && (k >> 6) < ((P.K + 63) >> 6)) epart[row * NW + (k >> 6)] = sum;
}

// The device noise as a kernel of its own, one lane per (sample, step pair): materialises the
// noise of a rollout that did not store it (mppi_download_noise, or a separate mppi_update after a
// tick).  (Drawing the noise first and rolling out on the stored eps was tried as a small-K tick
// path: slower than the fused rollout at every K from 1e4 to 5e5.)   grid = (ceil(K/256), ceil(T / steps per draw), A)
template <typename S, int PACK = 0>
__global__ __launch_bounds__(256) void eps_regen_kernel(DevParams P, S* __restrict__ eps, uint64_t seed, uint32_t tick_arg,
                                                       const uint32_t* __restrict__ tick_ptr) {
    constexpr int kS = NoisePack<PACK>::kSteps;
    const int k = blockIdx.x * 256 + threadIdx.x, triple = blockIdx.y, a = blockIdx.z;
    if (k >= P.K) return;
    const uint32_t tick = tick_ptr ? *tick_ptr : tick_arg;
    float e[2 * kS];
    philox_normals<PACK>(P.sample_offset + (uint32_t)k, (uint32_t)triple, tick, P.agent_offset + (uint32_t)a, (uint32_t)seed,
                         (uint32_t)(seed >> 32), (float)P.sigma, e);
    const size_t Ks = (size_t)P.Ks;
#pragma unroll
    for (int i = 0; i < kS; ++i) {
        const int t = kS * triple + i;
        if (t < P.T) {
            S* row = eps + (((size_t)a * P.T + t) * 2) * Ks + k;
            row[0] = (S)e[2 * i]; row[Ks] = (S)e[2 * i + 1];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// tuple merge:  M = min m_i;  D = sum exp(-(m_i - M)/lam) D_i  (same for N);
// E, count add.  Exact algebra of splitting the K-sum of control/src/mppi:189-196.
// merge_kernel: one wave per (t, a) reduces the NCH chunk tuples of this shard.
// ---------------------------------------------------------------------------------------------
#ifndef MPPI_ROLLOUT_TU  // non-template kernels are emitted by the engine translation unit only
__global__ __launch_bounds__(256) void merge_kernel(DevParams P, const double* __restrict__ part, int NCH,
                                                   double* __restrict__ merged) {
    // block = 64 threads (few chunk tuples) or 256 (the scan kernel's many block tuples)
    __shared__ double sh[4][7];
    const int t = blockIdx.x, a = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nw = (int)blockDim.x >> 6;
    const double* src = part + ((size_t)a * P.T + t) * NCH * kTupleW;
    // Up to kR tuples per thread are merged from registers: every one of them requested before anything is waited for (a clamped index
    // instead of a per-lane guard around the loads -- a guard gives each load its own basic block and its own wait), one round trip to
    // memory instead of three (count word -> minimum -> the rest).  More tuples than that: the two-pass loops.
    constexpr int kR = 4;
    const bool in_regs = NCH <= kR * (int)blockDim.x;   // (uniform)
    double q[kR][7];
    bool has[kR];
    double m = INFINITY;
    if (in_regs) {
#pragma unroll
        for (int r = 0; r < kR; ++r) {
            const int i = tid + r * (int)blockDim.x;
            const double* p = src + (size_t)min(i, NCH - 1) * kTupleW;
#pragma unroll
            for (int c = 0; c < 7; ++c) q[r][c] = p[c];
        }
#pragma unroll
        for (int r = 0; r < kR; ++r) {
            has[r] = tid + r * (int)blockDim.x < NCH && q[r][6] > 0.0;
            if (has[r]) m = fmin(m, q[r][0]);
        }
    } else {
        for (int i = tid; i < NCH; i += (int)blockDim.x)
            if (src[i * kTupleW + 6] > 0.0) m = fmin(m, src[i * kTupleW]);
    }
    double M = wave_min(m);
    if (nw > 1) {
        if (lane == 0) sh[wid][0] = M;
        __syncthreads();
        M = sh[0][0];
        for (int w = 1; w < nw; ++w) M = fmin(M, sh[w][0]);
        __syncthreads();
    }
    double d = 0, n0 = 0, n1 = 0, e0 = 0, e1 = 0, cnt = 0;
    if (in_regs) {
#pragma unroll
        for (int r = 0; r < kR; ++r)
            if (has[r]) {
                const double sc = exp((M - q[r][0]) * P.inv_lambda);
                d += sc * q[r][1]; n0 += sc * q[r][2]; n1 += sc * q[r][3]; e0 += q[r][4]; e1 += q[r][5]; cnt += q[r][6];
            }
    } else {
        for (int i = tid; i < NCH; i += (int)blockDim.x) {
            const double* qq = src + i * kTupleW;
            if (qq[6] > 0.0) {
                const double sc = exp((M - qq[0]) * P.inv_lambda);
                d += sc * qq[1]; n0 += sc * qq[2]; n1 += sc * qq[3]; e0 += qq[4]; e1 += qq[5]; cnt += qq[6];
            }
        }
    }
    d = wave_sum(d); n0 = wave_sum(n0); n1 = wave_sum(n1);
    e0 = wave_sum(e0); e1 = wave_sum(e1); cnt = wave_sum(cnt);
    if (nw > 1) {
        if (lane == 0) { sh[wid][1] = d; sh[wid][2] = n0; sh[wid][3] = n1; sh[wid][4] = e0; sh[wid][5] = e1; sh[wid][6] = cnt; }
        __syncthreads();
        if (tid == 0)
            for (int w = 1; w < nw; ++w) { d += sh[w][1]; n0 += sh[w][2]; n1 += sh[w][3]; e0 += sh[w][4]; e1 += sh[w][5]; cnt += sh[w][6]; }
    }
    if (tid == 0) {
        double* o = merged + ((size_t)a * P.T + t) * kTupleW;
        o[0] = M; o[1] = d; o[2] = n0; o[3] = n1; o[4] = e0; o[5] = e1; o[6] = cnt; o[7] = 0.0;
    }
}
#endif

// ---------------------------------------------------------------------------------------------
// scan_tick_kernel: rollout + cost-to-go + softmax partials of one tick in ONE kernel for small
// K (BASELINE config 2, the node's own K = 10): lanes = TIMESTEPS, one wave (T <= 64) or one block of
// four waves (T <= 256) per sample.  With constant wheel speeds over a step, theta is a prefix sum of
// the per-step rotations and (x, y) prefix sums of the per-step RK4 increments (which need only theta
// at the step's start), so a sample's whole trajectory is three scans deep instead of T steps long;
// the cost-to-go V[t] = sum_{tau >= t} c[tau] is a fourth.  The lane of timestep t then folds the
// sample into ITS running softmax tuple {min V, sum e, sum e*eps0, sum e*eps1, sum eps0, sum eps1,
// count} (an online softmax over the samples this unit walks through) -- V and eps never leave
// registers.  Same formulas as nominal_lanes (which is this with eps = 0); the noise is the same
// Philox stream, addressed by (sample, t / 3), so the result equals the lane-per-sample path's to
// rounding.  grid = (blocks, A) x 256; unit = wave (NWAVES 1: 4 units per block) or block (NWAVES 4);
// unit u walks samples [u * spw, (u + 1) * spw).  part[a][t][block] = the block's tuple; block 0 also
// snapshots (state, goal, nominal) so that V can be materialised later (mppi_download_value).
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
template <typename S, int NWAVES, bool PHILOX>
__global__ __launch_bounds__(256) void scan_tick_kernel(DevParams P, const double* __restrict__ state,
                                                       const double* __restrict__ goal,
                                                       const double* __restrict__ unom, const S* __restrict__ eps,
                                                       uint64_t seed, uint32_t tick_arg,
                                                       const uint32_t* __restrict__ tick_ptr, int spw,
                                                       double* __restrict__ part, int NB, double* __restrict__ prev,
                                                       double* __restrict__ state_keep, double* __restrict__ goal_keep) {
    __shared__ double sh_scan[4];
    __shared__ double sh_tup[NWAVES == 1 ? 4 * 64 * 7 : 1];
    const int tid = threadIdx.x, a = blockIdx.y, T = P.T, K = P.K;
    const int lane = tid & 63, wid = tid >> 6;
    const int t = NWAVES == 1 ? lane : tid;  // this thread's timestep
    const bool valid = t < T;
    const int unit = NWAVES == 1 ? (int)blockIdx.x * 4 + wid : (int)blockIdx.x;
    const bool first_block = blockIdx.x == 0;
    const double x0 = state[a * 3 + 0], y0 = state[a * 3 + 1], th0 = state[a * 3 + 2];
    const double gx = goal[a * 3 + 0], gy = goal[a * 3 + 1], gth = goal[a * 3 + 2];
    const double un0 = valid ? unom[(a * 2 + 0) * T + t] : 0.0;
    const double un1 = valid ? unom[(a * 2 + 1) * T + t] : 0.0;
    const uint32_t tick = PHILOX ? (tick_ptr ? *tick_ptr : tick_arg) : 0u;   // (with the other inputs: one round trip, not one more behind the snapshot)
    if (first_block) {  // pre-tick snapshot: prev = {unom [A][2][T], state [A][3], goal [A][3]}
        double* pv_state = prev + (size_t)P.A * 2 * T;
        double* pv_goal = pv_state + (size_t)P.A * 3;
        if ((NWAVES == 4 || wid == 0) && valid) { prev[(a * 2 + 0) * T + t] = un0; prev[(a * 2 + 1) * T + t] = un1; }
        if (tid < 3) {
            // (from the values read above, not from memory again: `state` / `goal` may be the caller's pinned host slot -- zero-copy
            // input: no H2D copy in front of the tick -- and every further load of it is another trip over PCIe in front of this block)
            const double sv = tid == 0 ? x0 : (tid == 1 ? y0 : th0), gv = tid == 0 ? gx : (tid == 1 ? gy : gth);
            pv_state[a * 3 + tid] = sv; pv_goal[a * 3 + tid] = gv;
            // the device-resident copies the later kernels (finalize's plant step) read are refreshed here
            if (state_keep != state) state_keep[a * 3 + tid] = sv;
            if (goal_keep != goal) goal_keep[a * 3 + tid] = gv;
        }
    }
    const double half_uru = 0.5 * (P.r0 * un0 * un0 + P.r1 * un1 * un1 + (P.offdiag ? 2.0 * (P.r01 * un0 * un1) : 0.0));
    double w0, w1;
    cost_noise_weights(P, un0, un1, w0, w1);
    double sth0, cth0;
    sincos(th0, &sth0, &cth0);
    uint32_t key0 = 0, key1 = 0;
    float sigf = 0.f;
    if (PHILOX) {
        key0 = (uint32_t)seed; key1 = (uint32_t)(seed >> 32);
        sigf = (float)P.sigma;
    }
    double m = INFINITY, D = 0.0, N0 = 0.0, N1 = 0.0, E0 = 0.0, E1 = 0.0, cnt = 0.0;
    const int k_lo = unit * spw, k_hi = min(K, k_lo + spw);  // uniform per unit (NWAVES 4: per block)
    for (int k = k_lo; k < k_hi; ++k) {
        double e0 = 0.0, e1 = 0.0;
        if (valid) {
            if (PHILOX) {
                float f0, f1;  // only this step's pair of the draw's three
                philox_normal_pair(P.sample_offset + (uint32_t)k, (uint32_t)t, tick, P.agent_offset + (uint32_t)a, key0, key1, sigf, f0, f1);
                e0 = (double)(S)f0;
                e1 = (double)(S)f1;
            } else {
                const S* ep = eps + ((size_t)a * T + t) * 2 * (size_t)P.Ks + k;
                e0 = (double)ep[0];
                e1 = (double)ep[(size_t)P.Ks];
            }
        }
        // EXPLORE + CLIP (control/src/mppi:147-152), then rk4 (:39-54) / euler (:57-58) as scans over t
        const double u0 = clampd(un0 + e0, P.u_max), u1 = clampd(un1 + e1, P.u_max);
        const double h = valid ? (P.model == 1 ? P.dt * u1 : P.kth * P.dt * (u1 - u0)) : 0.0;
        double tot_;
        const double th = th0 + (lanes_scan_incl<NWAVES>(h, t, sh_scan, tot_) - h);
        double s0, c0, ix, iy;
        if (NWAVES == 1 && P.model == 0 && __all(fabs(h) <= 0.5)) {
            // headings without a sincos per lane: every step is a small rotation (series), the heading at the
            // END of step t is the running product of the rotations (lane 0's carries the initial heading), the
            // heading at its start is that product turned back by the step's own rotation
            double sp, cp;
            small_sincos<7>(0.5 * h, sp, cp);
            const double hc = fma(-2.0 * sp, sp, 1.0), hs = 2.0 * sp * cp;  // (cos h, sin h)
            double c2 = lane == 0 ? hc * cth0 - hs * sth0 : hc, s2 = lane == 0 ? hs * cth0 + hc * sth0 : hs;
            wave_scan_rotations(c2, s2);
            c0 = c2 * hc + s2 * hs; s0 = s2 * hc - c2 * hs;
            const double c1 = c0 * cp - s0 * sp, s1 = s0 * cp + c0 * sp;
            const double aa = P.dt * P.rhalf * (u0 + u1) * (1.0 / 6.0);
            ix = aa * (c0 + 4.0 * c1 + c2); iy = aa * (s0 + 4.0 * s1 + s2);
        } else {
        sincos(th, &s0, &c0);
        if (P.model == 1) {
            ix = P.dt * (c0 * u0); iy = P.dt * (s0 * u0);
        } else {
            double s1, c1, s2, c2;
            if (fabs(h) <= 0.5) {
                double sp, cp;
                small_sincos<7>(0.5 * h, sp, cp);
                c1 = c0 * cp - s0 * sp; s1 = s0 * cp + c0 * sp;
                c2 = c1 * cp - s1 * sp; s2 = s1 * cp + c1 * sp;
            } else {
                sincos(th + 0.5 * h, &s1, &c1);
                sincos(th + h, &s2, &c2);
            }
            const double aa = P.dt * P.rhalf * (u0 + u1) * (1.0 / 6.0);
            ix = aa * (c0 + 4.0 * c1 + c2); iy = aa * (s0 + 4.0 * s1 + s2);
        }
        }
        const double X = x0 + lanes_scan_incl<NWAVES>(valid ? ix : 0.0, t, sh_scan, tot_);
        const double Y = y0 + lanes_scan_incl<NWAVES>(valid ? iy : 0.0, t, sh_scan, tot_);
        double cst = 0.0;
        if (valid) {  // get_cost (:180-184): u = NOMINAL, eps = UNCLIPPED; terminal cost (:165-173) at T-1
            const double the = th + h;  // the wrap (:52-53) is the identity on (-pi, pi]
            const double thn = (P.model == 0 && (the > M_PI || the <= -M_PI)) ? wrap_theta(the) : the;
            const double dx = X - gx, dy = Y - gy, dth = thn - gth;
            cst = 0.5 * (P.q0 * dx * dx + P.q1 * dy * dy + P.q2 * dth * dth) + half_uru + (w0 * e0 + w1 * e1);
            if (P.offdiag) cst += 0.5 * cross3(P.q01, P.q02, P.q12, dx, dy, dth);
            if (P.grid_weight != 0.0) cst += obstacle_cost(P, X, Y);
            if (t == T - 1) {
                cst += P.p0 * dx * dx + P.p1 * dy * dy + P.p2 * dth * dth;
                if (P.offdiag) cst += cross3(P.p01, P.p02, P.p12, dx, dy, dth);
            }
        }
        double tot;
        const double inc = lanes_scan_incl<NWAVES>(cst, t, sh_scan, tot);
        const double V = tot - (inc - cst);  // value_fcn[t, k] (:175)
        if (valid) {  // fold the sample into this timestep's tuple (update_action :187-196, online form)
            // ONE exponential per sample and step, no divergence: w = exp(-|V - m| / lam) either rescales the running
            // tuple (V is the new minimum; the sample enters with weight 1) or is the sample's own weight.  The first
            // sample meets m = +inf: w = exp(-inf) = 0.  fp32-storage mode takes the weight from v_exp_f32 like the
            // update kernel does (relative error ~1e-7 of the weight), the fp64 mode from exp().
            const bool lt = V < m;
            const double arg = -fabs(V - m) * P.inv_lambda;
            const double w = sizeof(S) == 4 ? (double)__builtin_amdgcn_exp2f((float)(arg * 1.4426950408889634)) : exp(arg);
            D = lt ? fma(D, w, 1.0) : D + w;
            N0 = lt ? fma(N0, w, e0) : fma(w, e0, N0);
            N1 = lt ? fma(N1, w, e1) : fma(w, e1, N1);
            m = lt ? V : m;
            E0 += e0; E1 += e1; cnt += 1.0;
        }
    }
    if (NWAVES == 1) {  // the block's four waves hold four tuples per timestep: merge them (exact rescaling)
        double* mine = sh_tup + (wid * 64 + lane) * 7;
        mine[0] = m; mine[1] = D; mine[2] = N0; mine[3] = N1; mine[4] = E0; mine[5] = E1; mine[6] = cnt;
        __syncthreads();
        if (wid != 0) return;
        double M = INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const double* q = sh_tup + (w * 64 + lane) * 7; if (q[6] > 0.0) M = fmin(M, q[0]); }
        D = N0 = N1 = E0 = E1 = cnt = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const double* q = sh_tup + (w * 64 + lane) * 7;
            if (q[6] > 0.0) {
                const double sc = (q[0] == M) ? 1.0 : exp((M - q[0]) * P.inv_lambda);
                D += sc * q[1]; N0 += sc * q[2]; N1 += sc * q[3]; E0 += q[4]; E1 += q[5]; cnt += q[6];
            }
        }
        m = M;
    }
    if (valid) {
        double* o = part + (((size_t)a * T + t) * NB + blockIdx.x) * kTupleW;
        o[0] = m; o[1] = D; o[2] = N0; o[3] = N1; o[4] = E0; o[5] = E1; o[6] = cnt; o[7] = 0.0;
    }
}

// exact rk4 step in the reference's operation order (control/src/mppi:39-54), used for the plant
__device__ __forceinline__ void rk4_exact(const DevParams& P, const double x0[3], double u0, double u1,
                                          double out[3]) {
    const double sum = u0 + u1, om = P.kth * (u1 - u0);
    double k1[3], k2[3], k3[3], k4[3];
    k1[0] = P.dt * (P.rhalf * cos(x0[2]) * sum); k1[1] = P.dt * (P.rhalf * sin(x0[2]) * sum); k1[2] = P.dt * om;
    double th = x0[2] + k1[2] / 2;
    k2[0] = P.dt * (P.rhalf * cos(th) * sum); k2[1] = P.dt * (P.rhalf * sin(th) * sum); k2[2] = P.dt * om;
    th = x0[2] + k2[2] / 2;
    k3[0] = P.dt * (P.rhalf * cos(th) * sum); k3[1] = P.dt * (P.rhalf * sin(th) * sum); k3[2] = P.dt * om;
    th = x0[2] + k3[2];
    k4[0] = P.dt * (P.rhalf * cos(th) * sum); k4[1] = P.dt * (P.rhalf * sin(th) * sum); k4[2] = P.dt * om;
#pragma unroll
    for (int i = 0; i < 3; ++i) out[i] = x0[i] + (1.0 / 6.0) * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
    out[2] = wrap_theta(out[2]);
}

// ---------------------------------------------------------------------------------------------
// finalize_kernel: the tail of update_action (control/src/mppi:196-208) and of get_path
// (:91-101) for one agent per block.
//   gathered [G][A][T][8] shard partials (G = 1: this engine's own)
//   flags: bit5 (with bits 0 and 1) the NEXT tick's nominal table (nominal_table_lanes) from the pose and controls this tick leaves;
//          bit0 plant step (perform_action :210-213), bit1 receding-horizon shift (:100-101),
//          bit2 bump the device tick counter (graph replay), bit3 the filter's basis staged in LDS (4*(T-1) more doubles),
//          bit4 set the device tick counter to tick_set (eager ticks: the id after the one just run, so a
//          later mppi_tick_graph continues the stream instead of re-drawing it)
//   ufilt [A][2][T] filtered controls (un-shifted), outv [A][8] = {next_state[3], u_applied[2]}
// dynamic LDS = 4*T doubles (+ T*T with bit3).
// ---------------------------------------------------------------------------------------------
// One-shot peer-to-peer all-gather of the shard tuples (K sharded over the GPUs of one node) without a
// collective launch: every rank owns a MAILBOX in fine-grained device memory that its peers map through HIP IPC,
//     data [2][G][n] float64   (double-buffered by the parity of the tick's epoch; slot g = rank g's tuples)
//     flag [2][G] uint32, one 64-byte line each (the epoch of the tick whose tuples the slot holds)
// p2p_publish_kernel (one block per destination) stores this rank's tuples into slot `rank` of every peer's mailbox
// over xGMI with system-scope write-through stores, fences, and raises the flag; the finalize kernel of each rank
// polls its OWN mailbox's G flags (relaxed system-scope loads, one lane each), acquires, and merges.  Double
// buffering is enough: a peer can only publish epoch e + 2 after its finalize of e + 1, which needed this rank's
// e + 1 tuples, which this rank published after its own finalize of e had read buffer (e & 1).
struct P2PWait {  // finalize side: nullptr flags = no wait
    const uint32_t* flags;   // this rank's flag[parity][0..n): kFlagStride uint32 (64 B) apart
    int n;
    uint32_t epoch;
    unsigned long long timeout_ticks;  // of wall_clock64() (100 MHz); 0 = wait forever
};
constexpr int kFlagStride = 16;  // uint32 per flag line

#ifndef MPPI_ROLLOUT_TU  // non-template kernels are emitted by the engine translation unit only
struct P2PPeers { double* data[8]; uint32_t* flag[8]; };  // slot `rank` of each peer's mailbox, for one parity
__global__ __launch_bounds__(256) void p2p_publish_kernel(const double* __restrict__ src, int n, P2PPeers peers, uint32_t epoch) {
    double* dst = peers.data[blockIdx.x];
    const int nt = (int)blockDim.x;
    for (int i0 = threadIdx.x; i0 < n; i0 += 4 * nt) {   // four words per thread requested before the first is stored (clamped index, no guard)
        double w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = src[min(i0 + j * nt, n - 1)];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i0 + j * nt < n) __hip_atomic_store(dst + i0 + j * nt, w[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // system scope: every store of this wave has left for the peer
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(peers.flag[blockIdx.x], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// The same with the shard's merge folded in: with a handful of chunk (or scan-block) tuples per row the merge kernel is
// skipped and every publishing block merges the n_rows = A * T rows itself (redundantly per destination, 16 lanes per
// row: merge_row16) straight into its peer's mailbox -- one launch and one boundary less per tick.  (Not beyond 16 per row:
// round 3 let every lane fold up to four tuples first -- 64 per row -- and the two publishing blocks of a 51-chunk shard took
// 7 us longer than the 50-block merge launch they replaced: 141.4 against 134.2 us per co-scheduled tick.)
__global__ __launch_bounds__(256) void p2p_publish_merge_kernel(DevParams P, const double* __restrict__ part, int NCH, int n_rows,
                                                               P2PPeers peers, uint32_t epoch) {
    double* dst = peers.data[blockIdx.x];
    const int r = threadIdx.x >> 4, g = threadIdx.x & 15;   // 16 lanes per row, 16 rows per pass
    // Four passes' tuples (64 rows: the node's [T = 50][8] block whole) are requested before the first is merged: every round trip to
    // the tuples -- they come from the update kernel's write-through stores, i.e. from memory -- is a microsecond in front of the
    // peers' finalize kernels.  (Until round 5: a pass at a time, and within a pass the count word first and the rest behind it: eight
    // dependent round trips for T = 50.  A 1024-thread workgroup does the same in one pass but has to find a whole free CU: measured
    // slower next to another process's kernels.)
    constexpr int kB = 4;
    for (int r0 = 0; r0 < n_rows; r0 += 16 * kB) {          // (uniform trip count)
        double w[kB][7];
        bool mine[kB];
#pragma unroll
        for (int b = 0; b < kB; ++b) {
            const int row = r0 + b * 16 + r;
            mine[b] = row < n_rows && g < NCH;
            const double* p = mine[b] ? part + ((size_t)row * NCH + g) * kTupleW : part;
#pragma unroll
            for (int i = 0; i < 7; ++i) w[b][i] = p[i];
        }
#pragma unroll
        for (int b = 0; b < kB; ++b) {
            const int row = r0 + b * 16 + r;
            if (r0 + b * 16 >= n_rows) break;               // (uniform)
            double t[7];
            merge_row16_words(mine[b], w[b], P.inv_lambda, t);
            if (row < n_rows && g < kTupleW)   // lanes 0..7 of the row store the eight words of its tuple
                __hip_atomic_store(dst + (size_t)row * kTupleW + g, g < 7 ? t[g] : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(peers.flag[blockIdx.x], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
#endif

// The consumer side of the exchange: wait (bounded) until the n flags of this rank's mailbox carry `epoch`, then make the
// peers' stores visible to this CU.  Returns false when a flag did not arrive in time.  All threads of the block call it.
__device__ __forceinline__ bool p2p_wait_block(const P2PWait& wait) {
    __shared__ int p2p_late;
    if (threadIdx.x == 0) p2p_late = 0;
    __syncthreads();
    if ((int)threadIdx.x < wait.n) {
        const uint32_t* f = wait.flags + (size_t)threadIdx.x * kFlagStride;
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != wait.epoch) {
            if (wait.timeout_ticks && wall_clock64() - t0 > wait.timeout_ticks) { p2p_late = 1; break; }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // system scope: drop whatever this CU cached of the mailbox
    return p2p_late == 0;
}

#ifndef MPPI_ROLLOUT_TU
// self-test consumer: the same wait the finalize kernel does, then the G slots copied out for the host to check
__global__ __launch_bounds__(256) void p2p_check_kernel(P2PWait wait, const double* __restrict__ slots, int n_per_slot,
                                                       int slot_stride, double* __restrict__ out, int* __restrict__ status) {
    const bool ok = p2p_wait_block(wait);
    if (threadIdx.x == 0) *status = ok ? 0 : 1;
    if (!ok) return;
    for (int g = 0; g < wait.n; ++g)
        for (int i = threadIdx.x; i < n_per_slot; i += 256) out[(size_t)g * n_per_slot + i] = slots[(size_t)g * slot_stride + i];
}
#endif

// Where the G tuples of (agent a, row t) sit: element offsets of tuple (g, a, t) = g * gs + a * as + t * ts.
//   shard tuples after an exchange  [G][A][T][8]:       gs = A*T*8, as = T*8,    ts = 8
//   the scan kernel's block tuples  part[A][T][NB][8]:  gs = 8,     as = T*NB*8, ts = NB*8   (G = NB: no merge kernel)
struct TupleLayout { unsigned gs, as, ts; };

// finalize_block: the body of finalize_kernel for one agent `a`, run by all threads of the block.
// smem: 4*T doubles (+ T*T with flags bit3), 16-byte aligned.
__device__ __forceinline__ void finalize_block(const DevParams& P, int a, const double* gathered, int G, TupleLayout lay,
                                               const double* __restrict__ Smat, double* unom, double* ufilt, double* state,
                                               double* outv, uint32_t* tick_ptr, int flags, uint32_t tick_set,
                                               double* host_out, uint32_t* host_seq, uint32_t seq, char* smem_raw,
                                               P2PWait wait, const double* __restrict__ goal = nullptr, double* __restrict__ tc = nullptr,
                                               double* __restrict__ base = nullptr, PkRow* __restrict__ pk = nullptr,
                                               const double* state_src = nullptr, double* goal_keep = nullptr) {
    // state_src: where this tick's pose is read from when it is not `state` yet -- a tick whose rollout read its inputs straight from
    // the caller's pinned slot left them in the pre-tick snapshot (snapshot_inputs); `state` (device-resident) receives the pose the
    // plant step predicts either way.  goal_keep: the device-resident goal, refreshed from `goal` (the snapshot's) on the same occasion.
    if (state_src == nullptr) state_src = state;
    FinProbe fprobe(P, a);   // (diagnostic builds: stamps 16 ... of the probe buffer; nothing in the product)
    if (wait.flags) {  // peer-to-peer exchange: `gathered` is this rank's mailbox; wait until every peer's tuples are in
        if (!p2p_wait_block(wait)) {  // a peer never delivered: poison the outputs (mppi_get_outputs reports MPPI_E_TIMEOUT), touch nothing else
            if (threadIdx.x == 0) {
                outv[(size_t)a * 8 + 7] = 1.0;
                if (host_out) {
                    __hip_atomic_store(host_out + (size_t)a * 8 + 7, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(host_seq + a, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            return;
        }
    }
    double* un = reinterpret_cast<double*>(smem_raw);  // [2][T] updated + clipped
    double* uf = un + 2 * P.T;                          // [2][T] filtered + clipped
    double* Sl = uf + 2 * P.T;                          // [4][T-1] + [4]: staged copy of the filter's basis (flags bit3)
    __shared__ double trig[3][2];
    __shared__ double nx_sh[3];     // the pose the plant step predicts (flags bit5: the next tick's table starts from it)
    __shared__ double tab_sh[16];   // scan scratch of nominal_table_lanes
    // flags bit 6 (engines that can run the fused fp64 tick, T <= 64): the largest merged sum of weights of this agent's rows goes out
    // next to the outputs (out[5]) -- what the engine reads from pinned memory to tell the regimes of the closed loop apart: under way
    // a row's sum is 1 ... 3 (the best sample and a neighbour or two), parked at the goal it is in the thousands
    __shared__ double dsum_sh[64];
    const int tid = threadIdx.x, T = P.T;
    const bool report_regime = (flags & 64) != 0 && T <= 64;
    // the filter operator does not depend on anything this kernel waits for: fetch it now, all loads in
    // flight at once, and read it from LDS when the updated controls are ready
    const bool staged = (flags & 8) != 0;
    const int nb = T - 1;  // basis length = the filter window
    if (staged)
        for (int i = tid; i < 4 * nb + 4; i += blockDim.x) Sl[i] = Smat[i];
    // omg = exp(-V/lam) + 1e-8, normalised; uvec += eps . omg   (control/src/mppi:193-196), then clip (:198-199)
    auto apply = [&](int t, double d, double n0, double n1, double e0, double e1, double cnt) {
        const double den = d + P.floor_w * cnt;
        if (report_regime) dsum_sh[t] = d;
        const double du0 = (n0 + P.floor_w * e0) / den, du1 = (n1 + P.floor_w * e1) / den;
        un[t] = clampd(unom[((size_t)a * 2 + 0) * T + t] + du0, P.u_max);
        un[T + t] = clampd(unom[((size_t)a * 2 + 1) * T + t] + du1, P.u_max);
    };
    fprobe.mark(0);   // behind the p2p wait and the issue of the basis loads
    if (G > 1 && G <= 16) {
        // several tuples per row (shards after an exchange, or a handful of chunk / scan-block tuples): 16 lanes per row
        const int rows_per_pass = (int)blockDim.x >> 4, r = tid >> 4, g = tid & 15;
        for (int t0 = 0; t0 < T; t0 += rows_per_pass) {  // (uniform trip count: every lane takes part in the DPP steps)
            const int t = t0 + r;
            const double* q = (t < T && g < G) ? gathered + (size_t)a * lay.as + (size_t)t * lay.ts + (size_t)g * lay.gs : nullptr;
            double m[7];
            merge_row16(q, gathered + (size_t)a * lay.as, P.inv_lambda, m);
            if (g == 0 && t < T) apply(t, m[1], m[2], m[3], m[4], m[5], m[6]);
        }
    } else {
        for (int t = tid; t < T; t += blockDim.x) {
            double M = INFINITY;
            const double* row = gathered + (size_t)a * lay.as + (size_t)t * lay.ts;
            for (int g = 0; g < G; ++g) {
                const double* q = row + (size_t)g * lay.gs;
                if (q[6] > 0.0) M = fmin(M, q[0]);
            }
            double d = 0, n0 = 0, n1 = 0, e0 = 0, e1 = 0, cnt = 0;
            for (int g = 0; g < G; ++g) {
                const double* q = row + (size_t)g * lay.gs;
                if (q[6] > 0.0) {
                    const double sc = (q[0] == M) ? 1.0 : exp((M - q[0]) * P.inv_lambda);
                    d += sc * q[1]; n0 += sc * q[2]; n1 += sc * q[3]; e0 += q[4]; e1 += q[5]; cnt += q[6];
                }
            }
            apply(t, d, n0, n1, e0, e1, cnt);
        }
    }
    fprobe.mark(1);   // tuples merged, controls updated (this thread's share)
    __syncthreads();
    // the stand-alone update (mppi_update: no plant step): the updated, clipped controls BEFORE the filter -- what the reference's
    // update_action leaves in its caller's uvec (control/src/mppi:196-199, in place; the filtered array it returns is a new one) --
    // behind the filtered ones, ufilt[A][2][T] | unfiltered [A][2][T]  (mppi_get_unfiltered)
    if (!(flags & 1))
        for (int j = tid; j < 2 * T; j += blockDim.x) ufilt[(size_t)P.A * 2 * T + (size_t)a * 2 * T + j] = un[j];
    if (report_regime && tid < 64) {   // (one wave; the sums were written in front of the barrier above)
        const double dm = -wave_min(tid < T ? -dsum_sh[tid] : 0.0);
        if (tid == 0) dsum_sh[0] = dm;   // (read by thread 0 behind the barriers below)
    }
    fprobe.mark(2);
    {   // savgol_filter (:202) as u @ S, clip (:205-206).  With window T-1 the operator has rank 8 (savgol.hpp):
        //   (u @ S)[j] = sum_d p_d(e_j) c_d[shift_j],   c_d[s] = sum_i p_d(i) u[i + s]
        // so per wheel eight dot products against the orthonormal basis -- one wave each, sixteen in all -- and four
        // multiply-adds per output, instead of 2 T^2 multiply-adds over an operator of 8 T^2 bytes.
        __shared__ double coef[2][2][4];
        const double* Pb = staged ? Sl : Smat;
        const int lane = tid & 63, wave = tid >> 6, nwaves = (int)blockDim.x >> 6;
        for (int c = wave; c < 16; c += nwaves) {   // c = (wheel, shift, degree); uniform per wave
            const int w = c >> 3, sh = (c >> 2) & 1, d = c & 3;
            const double* ur = un + w * T + sh;
            const double* pr = Pb + (size_t)d * nb;
            double acc = 0.0;
            for (int i = lane; i < nb; i += 64) acc = fma(pr[i], ur[i], acc);
            acc = wave_sum(acc);
            if (lane == 0) coef[w][sh][d] = acc;
        }
        __syncthreads();
        fprobe.mark(3);   // the sixteen filter coefficients
        // position inside its window (left window starts at 0, right at 1).  Odd window: the left one up to its centre.  Even
        // window (odd horizon, scipy >= 1.x semantics, savgol.hpp): the left one for j < nb/2, and the one interior sample
        // j = nb/2 is the RIGHT window's cubic at the half-integer position nb/2 - 1/2 -- the basis' four extra values
        const bool even = (nb & 1) == 0;
        const int last_left = even ? nb / 2 - 1 : (nb - 1) / 2;
        for (int idx = tid; idx < 2 * T; idx += blockDim.x) {
            const int w = idx >= T, j = idx - w * T;
            const int sh = j <= last_left ? 0 : 1, e = j - sh;
            const bool mid = even && j == nb / 2;
            double acc = 0.0;
#pragma unroll
            for (int d = 0; d < 4; ++d) acc = fma(mid ? Pb[(size_t)4 * nb + d] : Pb[(size_t)d * nb + e], coef[w][sh][d], acc);
            uf[idx] = clampd(acc, P.u_max);
        }
    }
    __syncthreads();
    fprobe.mark(4);   // filtered controls
    // perform_action (:210-213): the three distinct stage angles of rk4 evaluated by three lanes
    // (euler + unicycle only needs the first)
    const double sum = uf[0] + uf[T], om = (P.model == 1) ? uf[T] : P.kth * (uf[T] - uf[0]);
    const double th0 = state_src[a * 3 + 2], k_th = P.dt * om;
    if ((flags & 1) && tid < 3) {
        const double ang = (tid == 0) ? th0 : (tid == 1 ? th0 + k_th / 2 : th0 + k_th);
        trig[tid][0] = cos(ang);
        trig[tid][1] = sin(ang);
    }
    if (goal_keep != nullptr && goal_keep != goal && tid < 3) goal_keep[a * 3 + tid] = goal[a * 3 + tid];
    if (!(flags & 1) && state_src != state && tid < 3) state[a * 3 + tid] = state_src[a * 3 + tid];
    for (int j = tid; j < T; j += blockDim.x) {
        ufilt[((size_t)a * 2 + 0) * T + j] = uf[j];
        ufilt[((size_t)a * 2 + 1) * T + j] = uf[T + j];
        if (flags & 2) {  // shift left, the freed column takes uvec_init[:, 0] (:100-101)
            unom[((size_t)a * 2 + 0) * T + j] = (j + 1 < T) ? uf[j + 1] : P.shift_fill[a * 2 + 0];
            unom[((size_t)a * 2 + 1) * T + j] = (j + 1 < T) ? uf[T + j + 1] : P.shift_fill[a * 2 + 1];
        } else {
            unom[((size_t)a * 2 + 0) * T + j] = uf[j];
            unom[((size_t)a * 2 + 1) * T + j] = uf[T + j];
        }
    }
    __syncthreads();
    fprobe.mark(5);   // stage angles' cos / sin, controls written
    if (tid == 0) {
        if (flags & 1) {  // same operation order as rk4 (:39-54) with dd_dynamics (:23-30)
            const double x0[3] = {state_src[a * 3 + 0], state_src[a * 3 + 1], th0};
            double k1[3], k2[3], k3[3], k4[3], xn[3];
            if (P.model == 1) {  // euler (:57-58) over unicycle_dynamics (:33-36)
                xn[0] = x0[0] + P.dt * (trig[0][0] * uf[0]);
                xn[1] = x0[1] + P.dt * (trig[0][1] * uf[0]);
                xn[2] = x0[2] + P.dt * uf[T];
            } else {
            k1[0] = P.dt * (P.rhalf * trig[0][0] * sum); k1[1] = P.dt * (P.rhalf * trig[0][1] * sum); k1[2] = k_th;
            k2[0] = P.dt * (P.rhalf * trig[1][0] * sum); k2[1] = P.dt * (P.rhalf * trig[1][1] * sum); k2[2] = k_th;
            k3[0] = k2[0]; k3[1] = k2[1]; k3[2] = k_th;
            k4[0] = P.dt * (P.rhalf * trig[2][0] * sum); k4[1] = P.dt * (P.rhalf * trig[2][1] * sum); k4[2] = k_th;
#pragma unroll
            for (int i = 0; i < 3; ++i) xn[i] = x0[i] + (1.0 / 6.0) * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
            xn[2] = wrap_theta(xn[2]);
            }
            double* o = outv + (size_t)a * 8;
            o[0] = xn[0]; o[1] = xn[1]; o[2] = xn[2]; o[3] = uf[0]; o[4] = uf[T];
            o[5] = report_regime ? dsum_sh[0] : 0.0;
            state[a * 3 + 0] = xn[0]; state[a * 3 + 1] = xn[1]; state[a * 3 + 2] = xn[2];
            nx_sh[0] = xn[0]; nx_sh[1] = xn[1]; nx_sh[2] = xn[2];
            if (host_out) {
                // zero-copy output: the five results go straight into the caller's pinned host buffer, then the
                // sequence word (system-scope release) -- mppi_get_outputs polls that word instead of issuing a D2H copy
                double* ho = host_out + (size_t)a * 8;
#pragma unroll
                for (int i = 0; i < 6; ++i) __hip_atomic_store(ho + i, o[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(host_seq + a, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        if ((flags & 4) && a == 0 && tick_ptr) *tick_ptr = *tick_ptr + 1u;
        if ((flags & 16) && a == 0 && tick_ptr) *tick_ptr = tick_set;
        fprobe.mark(6);   // plant step, outputs on their way
    }
    if ((flags & 35) == 35) {
        // The next tick's nominal table, behind everything the caller waits for (the outputs above are on their way): lanes =
        // timesteps, from the pose just predicted and the controls just shifted -- the values this kernel stored, taken from LDS.  A
        // tick whose inputs are exactly these (no fresh pose, no mppi_set_nominal in between: the engine keeps track) launches the
        // rollout variant that LOADS the table; any other computes it in its prologue as before.
        __syncthreads();
        const bool valid = tid < T;
        const double un0 = !valid ? 0.0 : (tid + 1 < T ? uf[tid + 1] : P.shift_fill[a * 2 + 0]);
        const double un1 = !valid ? 0.0 : (tid + 1 < T ? uf[T + tid + 1] : P.shift_fill[a * 2 + 1]);
        const double gx = goal[a * 3 + 0], gy = goal[a * 3 + 1], gth = goal[a * 3 + 2];
        if (T <= 64) {
            if (tid < 64) nominal_table_lanes<1>(P, a, nx_sh[0], nx_sh[1], nx_sh[2], gx, gy, gth, un0, un1, tid, tab_sh, tc, base, pk);
        } else {   // (T <= 256 <= blockDim.x: every thread takes part in the scans' barriers)
            nominal_table_lanes<0>(P, a, nx_sh[0], nx_sh[1], nx_sh[2], gx, gy, gth, un0, un1, tid, tab_sh, tc, base, pk);
        }
    }
}

#ifndef MPPI_ROLLOUT_TU  // non-template kernels are emitted by the engine translation unit only
__global__ __launch_bounds__(1024) void finalize_kernel(DevParams P, const double* __restrict__ gathered, int G, TupleLayout lay,
                                                      const double* __restrict__ Smat, double* __restrict__ unom,
                                                      double* __restrict__ ufilt, double* __restrict__ state,
                                                      double* __restrict__ outv, uint32_t* tick_ptr, int flags,
                                                      uint32_t tick_set, double* host_out, uint32_t* host_seq, uint32_t seq,
                                                      P2PWait wait, const double* __restrict__ goal, double* __restrict__ tc,
                                                      double* __restrict__ base, PkRow* __restrict__ pk, const double* state_src,
                                                      double* goal_keep) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    finalize_block(P, blockIdx.x, gathered, G, lay, Smat, unom, ufilt, state, outv, tick_ptr, flags, tick_set, host_out, host_seq,
                   seq, smem_raw, wait, goal, tc, base, pk, state_src, goal_keep);
}
#endif

#ifndef MPPI_ROLLOUT_TU  // non-template kernels are emitted by the engine translation unit only
// state / goal from the caller's pinned, device-mapped staging slot into their device-resident homes: one tiny launch in
// front of a lane-per-sample tick instead of two H2D copies (the thousands of rollout blocks must not each read the slot
// over PCIe; the scan tick's few blocks do read it in place).  nullptr = keep what is resident.
__global__ void fetch_inputs_kernel(const double* __restrict__ src_state, const double* __restrict__ src_goal,
                                    double* __restrict__ state, double* __restrict__ goal, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        if (src_state) state[i] = src_state[i];
        if (src_goal) goal[i] = src_goal[i];
    }
}

// perform_action alone (control/src/mppi:210-213): next = rk4(state, unom[:,0])
__global__ void plant_kernel(DevParams P, const double* __restrict__ state, const double* __restrict__ unom,
                             double* __restrict__ outv) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= P.A) return;
    const double x0[3] = {state[a * 3 + 0], state[a * 3 + 1], state[a * 3 + 2]};
    const double u0 = unom[((size_t)a * 2 + 0) * P.T], u1 = unom[((size_t)a * 2 + 1) * P.T];
    double xn[3];
    if (P.model == 1) {  // euler (:57-58) over unicycle_dynamics (:33-36)
        xn[0] = x0[0] + P.dt * (cos(x0[2]) * u0);
        xn[1] = x0[1] + P.dt * (sin(x0[2]) * u0);
        xn[2] = x0[2] + P.dt * u1;
    } else {
        rk4_exact(P, x0, u0, u1, xn);
    }
    double* o = outv + (size_t)a * 8;
    o[0] = xn[0]; o[1] = xn[1]; o[2] = xn[2]; o[3] = u0; o[4] = u1;
}
#endif

// receding-horizon shift alone (control/src/mppi:100-101); one block per (agent,row) so the
// read of column j+1 and the write of column j cannot race across blocks
#ifndef MPPI_ROLLOUT_TU  // non-template kernels are emitted by the engine translation unit only
__global__ void shift_kernel(DevParams P, double* __restrict__ unom) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* row = reinterpret_cast<double*>(smem_raw);
    double* u = unom + (size_t)blockIdx.x * P.T;
    for (int j = threadIdx.x; j < P.T; j += blockDim.x) row[j] = u[j];
    __syncthreads();
    for (int j = threadIdx.x; j < P.T; j += blockDim.x) u[j] = (j + 1 < P.T) ? row[j + 1] : P.shift_fill[blockIdx.x];
}
#endif

// host <-> storage conversions (parity / compatibility paths, not on the tick path)
//   rows: n_rows rows of K elements; src pitch K (host layout), dst pitch Ks
template <typename S>
__global__ void pack_rows_kernel(const double* __restrict__ src, S* __restrict__ dst, int K, int Ks,
                                 const double* __restrict__ row_bias, int rows_per_bias) {
    const size_t row = blockIdx.y;
    const double b = row_bias ? row_bias[row / rows_per_bias] : 0.0;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < K; k += gridDim.x * blockDim.x)
        dst[row * Ks + k] = (S)(src[row * K + k] - b);
}
template <typename S>
__global__ void unpack_rows_kernel(const S* __restrict__ src, double* __restrict__ dst, int K, int Ks,
                                   const double* __restrict__ row_bias, int rows_per_bias) {
    const size_t row = blockIdx.y;
    const double b = row_bias ? row_bias[row / rows_per_bias] : 0.0;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < K; k += gridDim.x * blockDim.x)
        dst[row * K + k] = (double)src[row * Ks + k] + b;
}
// value_fcn <-> resident form.  unpack: V[a][t][k] = base[a][t] + Stot[a][k] - dP[a][t][k];
// pack (caller-made V): dP = -(V - base), Stot = 0.   grid = (blocks over k, A*T)
template <typename S>
__global__ void value_unpack_kernel(const S* __restrict__ dP, const S* __restrict__ Stot,
                                    const double* __restrict__ base, double* __restrict__ dst, int K, int Ks, int T) {
    const size_t row = blockIdx.y, a = row / T;
    const double b = base[row];
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < K; k += gridDim.x * blockDim.x)
        dst[row * K + k] = b + ((double)Stot[a * Ks + k] - (double)dP[row * Ks + k]);
}
template <typename S>
__global__ void value_pack_kernel(const double* __restrict__ src, const double* __restrict__ base,
                                  S* __restrict__ dP, S* __restrict__ Stot, int K, int Ks, int T) {
    const size_t row = blockIdx.y, a = row / T;
    const double b = base[row];
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < K; k += gridDim.x * blockDim.x) {
        dP[row * Ks + k] = (S)(b - src[row * K + k]);
        if (row % T == 0) Stot[a * Ks + k] = (S)0;
    }
}
// per-row minimum of a [rows][K] float64 array (mppi_upload_value picks it as the baseline)
#ifndef MPPI_ROLLOUT_TU  // non-template kernels are emitted by the engine translation unit only
__global__ __launch_bounds__(256) void row_min_kernel(const double* __restrict__ src, int K, double* __restrict__ out) {
    __shared__ double red[4];
    const size_t row = blockIdx.x;
    double m = INFINITY;
    for (int k = threadIdx.x; k < K; k += 256) m = fmin(m, src[row * K + k]);
    m = wave_min(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[row] = fmin(fmin(red[0], red[1]), fmin(red[2], red[3]));
}
#endif

}  // namespace mppi
