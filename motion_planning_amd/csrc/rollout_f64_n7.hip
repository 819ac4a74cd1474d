// rollout_kernel instantiations: storage double, heading-rotation series NTERM = 7 (see rollout_launch.hpp)
#define MPPI_ROLLOUT_TU 1
#include "rollout_launch.hpp"
namespace mppi { template hipError_t launch_rollout_typed<double, 7>(const RolloutArgs&); }
