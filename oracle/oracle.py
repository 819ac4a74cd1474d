"""ctypes front-end of the CPU oracle (oracle/mppi_oracle.c).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.  Nothing under motion_planning_amd/
does (tests/test_abi_cpu.py::test_product_never_touches_the_oracle enforces it).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmppi_oracle.so")

c_dp = C.POINTER(C.c_double)


class OrcParams(C.Structure):
    _fields_ = [("q", C.c_double * 3), ("r", C.c_double * 2), ("p1", C.c_double * 3),
                ("u_max", C.c_double), ("wheel_radius", C.c_double),
                ("wheel_base", C.c_double), ("floor_w", C.c_double), ("model", C.c_int),
                ("grid", C.c_void_p), ("grid_w", C.c_int), ("grid_h", C.c_int),
                ("grid_res", C.c_double), ("grid_ox", C.c_double), ("grid_oy", C.c_double),
                ("grid_weight", C.c_double), ("use_sig", C.c_int), ("sig", C.c_double * 4),
                ("shift_fill", C.c_double * 2),
                ("use_full", C.c_int), ("Qf", C.c_double * 9), ("Rf", C.c_double * 4), ("P1f", C.c_double * 9)]


def build(force=False):
    src = os.path.join(_HERE, "mppi_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libmppi_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_set_threads.restype = C.c_int
        _lib.orc_savgol_matrix.restype = C.c_int
    return _lib


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(c_dp)


def default_params():
    p = OrcParams()
    lib().orc_default_params(C.byref(p))
    return p


def set_obstacle_grid(params, cells, resolution, origin, weight):
    """cells [height][width] int8 (map::Grid export: 0 / 50 / 100); keeps the array alive on params."""
    cells = np.ascontiguousarray(cells, dtype=np.int8)
    params._grid_keepalive = cells
    params.grid = cells.ctypes.data
    params.grid_h, params.grid_w = cells.shape
    params.grid_res, params.grid_ox, params.grid_oy = float(resolution), float(origin[0]), float(origin[1])
    params.grid_weight = float(weight)
    return params


def set_sig_matrix(params, sig):
    """sig [2,2] as the reference's get_path accepts it: stage cost lam * u . sig . eps (control/src/mppi:184);
    the noise is still drawn with sig[0,0] (:143-146) -- pass that as the scalar `sigma`."""
    sig = np.asarray(sig, dtype=np.float64).reshape(2, 2)
    params.use_sig = 1
    params.sig[:] = [sig[0, 0], sig[0, 1], sig[1, 0], sig[1, 1]]
    return params


def set_weight_matrices(params, Q, R, P1):
    """Q [3,3], R [2,2], P1 [3,3] as the reference's attributes (control/src/mppi:69-73), multiplied whole (:168, :181-184)."""
    Q, R, P1 = (np.asarray(m, dtype=np.float64) for m in (Q, R, P1))
    assert Q.shape == (3, 3) and R.shape == (2, 2) and P1.shape == (3, 3)
    params.use_full = 1
    params.Qf[:] = list(Q.ravel())
    params.Rf[:] = list(R.ravel())
    params.P1f[:] = list(P1.ravel())
    params.q[:] = list(np.diag(Q)); params.r[:] = list(np.diag(R)); params.p1[:] = list(np.diag(P1))
    return params


def euler(x0, u, dt, params=None):
    p = params or default_params()
    out = np.zeros(3)
    lib().orc_euler(C.byref(p), _p(_d(x0)), _p(_d(u)), C.c_double(dt), _p(out))
    return out


def set_threads(n):
    return lib().orc_set_threads(C.c_int(int(n)))


def dd_dynamics(x, u, params=None):
    p = params or default_params()
    out = np.zeros(3)
    lib().orc_dd_dynamics(C.byref(p), _p(_d(x)), _p(_d(u)), _p(out))
    return out


def rk4(x0, u, dt, params=None):
    p = params or default_params()
    out = np.zeros(3)
    lib().orc_rk4(C.byref(p), _p(_d(x0)), _p(_d(u)), C.c_double(dt), _p(out))
    return out


def get_cost2go(state, uvec, goal, lam, sigma, eps, dt=None, params=None, want_costs=False):
    """eps [T,2,K] -> V [T,K] (and optionally stage costs [T,K], final states [K,3])."""
    p = params or default_params()
    eps = _d(eps)
    T, _, K = eps.shape
    uvec = _d(uvec)
    assert uvec.shape == (2, T)
    dt = 1.0 / T if dt is None else dt
    V = np.zeros((T, K))
    cost = np.zeros((T, K)) if want_costs else None
    xT = np.zeros((K, 3)) if want_costs else None
    lib().orc_get_cost2go(C.byref(p), K, T, C.c_double(dt), _p(_d(state)), _p(uvec), _p(_d(goal)),
                          C.c_double(lam), C.c_double(sigma), _p(eps), _p(V),
                          _p(cost) if want_costs else None, _p(xT) if want_costs else None)
    return (V, cost, xT) if want_costs else V


def savgol_matrix(T):
    S = np.zeros((T, T))
    rc = lib().orc_savgol_matrix(T, _p(S))
    if rc != 0:
        raise ValueError("savgol window T-1=%d must be odd and > 3" % (T - 1))
    return S


def update_action(uvec, eps, V, lam, S=None, params=None, want_inplace=False):
    """Returns the filtered controls [2,T] (the arguments are copied first).  want_inplace: also what the reference leaves in its
    caller's uvec and value_fcn (control/src/mppi:189, :196-199: it mutates both in place) -> (out, uvec_after, V_after)."""
    p = params or default_params()
    eps = _d(eps)
    T, _, K = eps.shape
    u = _d(uvec).copy()
    Vc = _d(V).copy()
    S = savgol_matrix(T) if S is None else _d(S)
    out = np.zeros((2, T))
    lib().orc_update_action(C.byref(p), K, T, _p(u), _p(eps), _p(Vc), C.c_double(lam), _p(S), _p(out))
    return (out, u, Vc) if want_inplace else out


def get_path(state, goal, latest_uvec, eps, lam=0.001, sigma=0.9, dt=None, S=None, params=None):
    """One tick.  Returns (next_state[3], u_applied[2], latest_uvec_after_shift[2,T])."""
    p = params or default_params()
    eps = _d(eps)
    T, _, K = eps.shape
    dt = 1.0 / T if dt is None else dt
    S = savgol_matrix(T) if S is None else _d(S)
    lat = _d(latest_uvec).copy()
    nxt = np.zeros(3)
    ua = np.zeros(2)
    lib().orc_get_path(C.byref(p), K, T, C.c_double(dt), _p(_d(state)), _p(_d(goal)),
                       C.c_double(lam), C.c_double(sigma), _p(eps), _p(S), _p(lat), _p(nxt), _p(ua))
    return nxt, ua, lat


def wheels_to_twist(u, params=None):
    p = params or default_params()
    out = np.zeros(2)
    lib().orc_wheels_to_twist(C.byref(p), _p(_d(u)), _p(out))
    return out


def shard_partials(eps, V, k0, k1, lam):
    eps = _d(eps)
    V = _d(V)
    T, _, K = eps.shape
    part = np.zeros((T, 6))
    lib().orc_shard_partials(K, T, int(k0), int(k1), _p(eps), _p(V), C.c_double(lam), _p(part))
    return part


def merge_partials(parts, counts, lam, floor_w=1e-8):
    parts = _d(parts)
    G, T, _ = parts.shape
    counts = _d(counts)
    du = np.zeros((2, T))
    lib().orc_merge_partials(G, T, _p(parts), _p(counts), C.c_double(lam), C.c_double(floor_w), _p(du))
    return du


def philox4x32_10(ctr, key):
    c = (C.c_uint32 * 4)(*[int(x) for x in ctr])
    k = (C.c_uint32 * 2)(*[int(x) for x in key])
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return [int(x) for x in o]


def philox_noise(seed, agent, tick, k_off, K_local, T, sigma, packing=0):
    """The device noise's CPU twin; packing = the engine's option "noise_packing" (0: three steps per Philox call, 1: four,
    2: hipRAND's own normals, two)."""
    eps = np.zeros((T, 2, K_local))
    fn = {0: lib().orc_philox_noise, 1: lib().orc_philox_noise16, 2: lib().orc_philox_noise_hiprand}[int(packing)]
    fn(C.c_uint64(int(seed)), C.c_uint32(int(agent)), C.c_uint32(int(tick)),
       C.c_uint32(int(k_off)), int(K_local), int(T), C.c_double(sigma), _p(eps))
    return eps


def reference_noise(seed, sigma, T, K, n_ticks=None):
    """The reference's noise stream (control/src/mppi:143-146 under np.random.seed(seed)):
    numpy's frozen legacy MT19937 stream, see tests/golden/make_golden.py."""
    rs = np.random.RandomState(seed)
    shape = (T, 2, K) if n_ticks is None else (n_ticks, T, 2, K)
    return rs.normal(0.0, sigma, shape)
