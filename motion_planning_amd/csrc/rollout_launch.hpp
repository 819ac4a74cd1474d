// rollout_launch.hpp -- host-side selection of the rollout_kernel instantiation.
// The kernel has seven template axes (storage, rotation series, noise source, eps store, inline
// nominal, model, cost terms); to keep the build parallel its instantiations live in one translation
// unit per (storage type, rotation series): rollout_f32_n4.hip ... rollout_f64_n0.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "mppi_kernels.hpp"

namespace mppi {

struct RolloutArgs {
    DevParams P;
    hipStream_t stream;
    int k0, k1;              // samples [k0, k1) of this launch
    bool philox, store_eps;  // noise source; whether device noise is written to HBM
    int inline_nominal;      // 0: table from nominal_kernel, 1: one wave (T <= 64), 2: four waves (T <= 256)
    int model;               // MPPI_MODEL_*
    bool general;            // Q[2,2] != 0 or an obstacle grid is set
    uint64_t seed;
    uint32_t tick;
    const uint32_t* tick_ptr;
    const double *state, *goal, *unom;
    double *tc, *base;
    void *eps, *dP, *stot, *epart;  // S-typed buffers
    // non-null: the launch itself carries the two events (hipExtLaunchKernelGGL) -- they take the
    // dispatch's own begin / end timestamps, the clock rocprofv3 --kernel-trace reads, and put no
    // marker packets into the stream
    hipEvent_t ev_start, ev_stop;
};

// returns hipGetLastError() of the launch
template <typename S, int NTERM>
hipError_t launch_rollout_typed(const RolloutArgs& a);

#ifdef MPPI_ROLLOUT_TU
// ---- body, compiled only inside the rollout_*.hip translation units ---------------------------------
template <typename S, int NT, bool PH, bool SE, int IN, int MODEL, bool GEN>
static hipError_t rollout_go(const RolloutArgs& a) {
    auto kern = rollout_kernel<S, NT, PH, SE, IN, MODEL, GEN>;
    dim3 grid((a.k1 - a.k0 + 255) / 256, a.P.A);
    const unsigned lds = (unsigned)((size_t)a.P.T * 5 * sizeof(double));
    if (a.ev_start)
        hipExtLaunchKernelGGL(kern, grid, dim3(256), lds, a.stream, a.ev_start, a.ev_stop, 0, a.P, a.state, a.goal, a.tc,
                              static_cast<S*>(a.eps), static_cast<S*>(a.dP), static_cast<S*>(a.stot), a.seed, a.tick,
                              a.tick_ptr, a.k0, a.k1, static_cast<S*>(a.epart), a.unom, a.base);
    else
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, a.stream, a.P, a.state, a.goal, a.tc,
                           static_cast<S*>(a.eps), static_cast<S*>(a.dP), static_cast<S*>(a.stot), a.seed, a.tick, a.tick_ptr,
                           a.k0, a.k1, static_cast<S*>(a.epart), a.unom, a.base);
    return hipGetLastError();
}
template <typename S, int NT, bool PH, bool SE, int IN, int MODEL>
static hipError_t rollout_gen(const RolloutArgs& a) {
    // the node's cost (Q[2,2] = 0, no obstacle grid) runs the branch-free instantiation
    return a.general ? rollout_go<S, NT, PH, SE, IN, MODEL, true>(a) : rollout_go<S, NT, PH, SE, IN, MODEL, false>(a);
}
template <typename S, int NT, bool PH, bool SE>
static hipError_t rollout_model(const RolloutArgs& a) {
    if (a.model == 1) return rollout_gen<S, NT, PH, SE, 0, 1>(a);
    if (a.inline_nominal == 1) return rollout_gen<S, NT, PH, SE, 1, 0>(a);
    if (a.inline_nominal == 2) return rollout_gen<S, NT, PH, SE, 2, 0>(a);
    return rollout_gen<S, NT, PH, SE, 0, 0>(a);
}
template <typename S, int NTERM>
hipError_t launch_rollout_typed(const RolloutArgs& a) {
    if (!a.philox) return rollout_model<S, NTERM, false, false>(a);
    return a.store_eps ? rollout_model<S, NTERM, true, true>(a) : rollout_model<S, NTERM, true, false>(a);
}
#endif

}  // namespace mppi
