// rollout_kernel instantiations: storage float, heading-rotation series NTERM = 4 (see rollout_launch.hpp)
#define MPPI_ROLLOUT_TU 1
#include "rollout_launch.hpp"
namespace mppi { template hipError_t launch_rollout_typed<float, 4>(const RolloutArgs&); }
