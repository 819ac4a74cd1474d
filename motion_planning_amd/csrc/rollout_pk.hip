// rollout_pk_kernel (mixed-precision, two samples per lane): see rollout_pk.hpp
#define MPPI_ROLLOUT_TU 1
#define MPPI_ROLLOUT_PK_TU 1
#include "rollout_pk.hpp"
