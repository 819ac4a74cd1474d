// tick_fused_kernel with the mixed-precision rollout (rollout_pk_body) as its rollout work items: see tick_fused.hpp
#define MPPI_ROLLOUT_TU 1
#define MPPI_FUSED_TU 1
#define MPPI_FUSED_PK_TU 1
#include "rollout_pk.hpp"
#include "tick_fused.hpp"
namespace mppi {
hipError_t launch_tick_fused_pk(const FusedLaunch& a) {
    return a.inline_nominal == 2 ? tick_fused_go<FusedRollPk<2>>(a) : tick_fused_go<FusedRollPk<1>>(a);
}
hipError_t launch_rollout_arrive_pk(const FusedLaunch& a) {
    return a.inline_nominal == 2 ? rollout_arrive_go<FusedRollPk<2>>(a) : rollout_arrive_go<FusedRollPk<1>>(a);
}
}  // namespace mppi
