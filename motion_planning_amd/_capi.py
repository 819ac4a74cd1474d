"""ctypes binding of include/mppi_hip.h and include/mppi_hip_diag.h (libmppi_hip.so).  This is the whole FFI: the
reference has none (its controller is a single Python script, control/src/mppi), so this
file is the binding INTEGRATION.md tells a maintainer to add."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmppi_hip.so")

MPPI_STORE_F32, MPPI_STORE_F64 = 0, 1
MPPI_NOISE_INJECTED, MPPI_NOISE_PHILOX = 0, 1
MPPI_MODEL_DIFFDRIVE_RK4, MPPI_MODEL_UNICYCLE_EULER = 0, 1
MPPI_TICK_AUTO, MPPI_TICK_LANES, MPPI_TICK_SCAN = 0, 1, 2
MPPI_E_TIMEOUT = -5
IPC_HANDLE_BYTES = 64
KERNELS = ("nominal", "rollout", "update", "merge", "finalize", "exchange")
ABI_VERSION = 5
PROBE_MARKS = 30


class MppiConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_agents", C.c_int32), ("samples", C.c_int32), ("horizon", C.c_int32),
                ("storage", C.c_int32), ("device", C.c_int32), ("sample_offset", C.c_uint32),
                ("model", C.c_int32), ("tick_path", C.c_int32), ("co_shards", C.c_int32), ("agent_offset", C.c_int32),
                ("reserved0", C.c_int32), ("dt", C.c_double), ("sigma", C.c_double), ("lambda_", C.c_double),
                ("q", C.c_double * 3), ("r", C.c_double * 2), ("p1", C.c_double * 3),
                ("u_max", C.c_double), ("wheel_radius", C.c_double), ("wheel_base", C.c_double),
                ("floor_w", C.c_double), ("samples_total", C.c_int64)]


class MppiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libmppi_hip error %d: %s" % (code, msg))
        self.code = code


_dp = C.POINTER(C.c_double)
_H = C.c_void_p

# name -> (restype, argtypes); every symbol include/mppi_hip.h and include/mppi_hip_diag.h declare
SIGNATURES = {
    "mppi_default_config": (C.c_int, [C.POINTER(MppiConfig)]),
    "mppi_abi_version": (C.c_int, []),
    "mppi_last_error": (C.c_char_p, [_H]),
    "mppi_create": (C.c_int, [C.POINTER(MppiConfig), C.POINTER(_H)]),
    "mppi_destroy": (C.c_int, [_H]),
    "mppi_set_stream": (C.c_int, [_H, C.c_void_p]),
    "mppi_get_stream": (C.c_int, [_H, C.POINTER(C.c_void_p)]),
    "mppi_set_sigma_lambda": (C.c_int, [_H, C.c_double, C.c_double]),
    "mppi_set_sig_matrix": (C.c_int, [_H, _dp, C.c_double]),
    "mppi_set_weights": (C.c_int, [_H, _dp, _dp, _dp]),
    "mppi_set_weight_matrices": (C.c_int, [_H, _dp, _dp, _dp]),
    "mppi_set_sync_timeout": (C.c_int, [_H, C.c_int]),
    "mppi_set_tick_counter": (C.c_int, [_H, C.c_uint32]),
    "mppi_stream_wait_partials": (C.c_int, [_H, C.c_void_p]),
    "mppi_wait_for_stream": (C.c_int, [_H, C.c_void_p]),
    "mppi_reset": (C.c_int, [_H, C.c_int]),
    "mppi_set_obstacle_grid": (C.c_int, [_H, C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double,
                                         C.c_double]),
    "mppi_set_shift_fill": (C.c_int, [_H, C.c_int, _dp]),
    "mppi_set_nominal": (C.c_int, [_H, C.c_int, _dp]),
    "mppi_get_nominal": (C.c_int, [_H, C.c_int, _dp]),
    "mppi_upload_noise": (C.c_int, [_H, _dp]),
    "mppi_download_noise": (C.c_int, [_H, _dp]),
    "mppi_rollout": (C.c_int, [_H, _dp, _dp, C.c_int, C.c_uint64, C.c_uint32]),
    "mppi_download_value": (C.c_int, [_H, _dp]),
    "mppi_upload_value": (C.c_int, [_H, _dp]),
    "mppi_update": (C.c_int, [_H, _dp]),
    "mppi_plant_step": (C.c_int, [_H, _dp, _dp]),
    "mppi_shift": (C.c_int, [_H]),
    "mppi_tick_begin": (C.c_int, [_H, _dp, _dp, C.c_int, C.c_uint64, C.c_uint32]),
    "mppi_partials_ptr": (C.c_int, [_H, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "mppi_tick_finish": (C.c_int, [_H, C.c_void_p, C.c_int]),
    "mppi_p2p_create": (C.c_int, [_H, C.c_int, C.c_int, C.c_void_p]),
    "mppi_p2p_connect": (C.c_int, [_H, C.c_void_p, C.POINTER(C.c_void_p)]),
    "mppi_p2p_rendezvous": (C.c_int, [_H, C.c_char_p, C.c_int, C.c_int, C.c_int]),
    "mppi_p2p_mailbox_ptr": (C.c_int, [_H, C.POINTER(C.c_void_p)]),
    "mppi_p2p_selftest": (C.c_int, [_H, C.c_int]),
    "mppi_p2p_destroy": (C.c_int, [_H]),
    "mppi_p2p_publish": (C.c_int, [_H]),
    "mppi_tick_finish_p2p": (C.c_int, [_H]),
    "mppi_tick_exchange_p2p": (C.c_int, [_H]),
    "mppi_get_outputs": (C.c_int, [_H, _dp, _dp]),
    "mppi_tick": (C.c_int, [_H, _dp, _dp, C.c_int, C.c_uint64, C.c_uint32, _dp, _dp]),
    "mppi_tick_graph": (C.c_int, [_H, C.c_uint64]),
    "mppi_synchronize": (C.c_int, [_H]),
    "mppi_get_unfiltered": (C.c_int, [_H, _dp]),
    "mppi_set_option": (C.c_int, [_H, C.c_char_p, C.c_int64]),
    "mppi_get_option": (C.c_int, [_H, C.c_char_p, C.POINTER(C.c_int64)]),
    "mppi_savgol_matrix": (C.c_int, [C.c_int, _dp]),
    "mppi_kernel_timing": (C.c_int, [_H, C.c_uint32]),
    "mppi_kernel_timing_period": (C.c_int, [_H, C.c_int]),
    "mppi_kernel_times": (C.c_int, [_H, _dp, C.POINTER(C.c_int64)]),
    "mppi_shader_clock": (C.c_int, [_H, _dp]),
    "mppi_probe_timeline": (C.c_int, [_H, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "mppi_co_info": (C.c_int, [_H, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mppi_co_note": (C.c_char_p, [_H]),
    "mppi_rollout_kernel": (C.c_int, [_H, C.POINTER(C.c_int32)]),
    "mppi_engine_info": (C.c_int, [_H, C.POINTER(C.c_size_t), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
}

_lib = None


def load():
    """Load libmppi_hip.so.  Fails loudly (no fallback) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C motion_planning_amd/csrc`.  motion_planning_amd has no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI drifted
        fn.restype = res
        fn.argtypes = args
    if lib.mppi_abi_version() != ABI_VERSION:
        raise ImportError("libmppi_hip ABI %d != binding %d" % (lib.mppi_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def default_config():
    cfg = MppiConfig()
    rc = load().mppi_default_config(C.byref(cfg))
    if rc:
        raise MppiError(rc, "mppi_default_config")
    return cfg


def check(rc, handle=None):
    if rc != 0:
        msg = load().mppi_last_error(handle)
        raise MppiError(rc, msg.decode() if msg else "?")


def dptr(arr):
    return arr.ctypes.data_as(_dp) if arr is not None else None
