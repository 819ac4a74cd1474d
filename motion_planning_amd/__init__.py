"""motion_planning_amd -- MI355X-native MPPI rollout engine.

Drop-in for the MPPI controller of moribots/motion_planning (`control/src/mppi`):
the K x T sampled-trajectory rollout, per-step cost, per-timestep softmax and weighted
control update run as hand-written gfx950 HIP kernels behind the C ABI of
``include/mppi_hip.h`` (``lib/libmppi_hip.so``).  There is no CPU fallback: importing
works anywhere, creating an engine needs the built library and a GPU.
"""
from . import _capi  # noqa: F401
from .mppi import MPPI, WHEEL_BASE, WHEEL_RADIUS, WHEEL_VEL_MAX, dd_dynamics, euler, rk4, unicycle_dynamics  # noqa: F401
from .controller import Controller, wheels_to_twist  # noqa: F401

__all__ = ["MPPI", "Controller", "rk4", "euler", "dd_dynamics", "unicycle_dynamics", "wheels_to_twist",
           "WHEEL_VEL_MAX", "WHEEL_RADIUS", "WHEEL_BASE"]
