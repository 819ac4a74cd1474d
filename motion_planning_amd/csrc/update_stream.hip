// update_stream_kernel: the update work items of a tick as persistent workgroups on a stream of their own (tick_fused.hpp)
#define MPPI_ROLLOUT_TU 1
#define MPPI_FUSED_TU 1
#define MPPI_UPDATE_STREAM_TU 1
#include "tick_fused.hpp"
