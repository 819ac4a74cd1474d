// tick_fused_kernel with the one-sample-per-lane rollout (rollout_body<float, NTERM = 0, ...>) as its rollout work items: see tick_fused.hpp
#define MPPI_ROLLOUT_TU 1
#define MPPI_FUSED_TU 1
#include "tick_fused.hpp"
namespace mppi {
template <> hipError_t launch_tick_fused_f32<0>(const FusedLaunch& a) {
    return a.inline_nominal == 2 ? tick_fused_go<FusedRollF32<0, 2>>(a) : tick_fused_go<FusedRollF32<0, 1>>(a);
}
template <> hipError_t launch_rollout_arrive_f32<0>(const FusedLaunch& a) {
    return a.inline_nominal == 2 ? rollout_arrive_go<FusedRollF32<0, 2>>(a) : rollout_arrive_go<FusedRollF32<0, 1>>(a);
}
}  // namespace mppi
