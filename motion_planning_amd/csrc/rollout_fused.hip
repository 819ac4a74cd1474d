// rollout_fused.hip -- the fused rollout + softmax-partials kernel of the fp64-storage tick (rollout_fused.hpp), a translation unit of its own
#define MPPI_ROLLOUT_TU 1
#define MPPI_ROLLOUT_FUSED_TU 1
#include "rollout_fused.hpp"
