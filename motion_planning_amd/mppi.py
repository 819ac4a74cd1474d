"""Host-side mirror of the reference's MPPI interface on top of libmppi_hip.so.

``Engine`` is the thin object wrapper over the C ABI (batched over agents, one GPU's
shard of the samples).  ``MPPI`` mirrors the reference class of the same name
(moribots/motion_planning ``control/src/mppi:61-213``): same constructor arguments, same
method names, argument meaning and array layouts, so the reference's ``Controller``
logic runs against it unmodified -- but every K x T loop runs in the HIP kernels.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import MPPI_NOISE_INJECTED, MPPI_NOISE_PHILOX, MPPI_STORE_F32, MPPI_STORE_F64

# control/src/mppi:18-20
WHEEL_VEL_MAX = 6.35492
WHEEL_RADIUS = 0.033
WHEEL_BASE = 0.16


def dd_dynamics(x, u):
    """Differential-drive kinematics, control/src/mppi:23-30: x = [x, y, theta] ([3] or [3, N]), u = wheel speeds
    [left, right] ([2] or [2, N]) -> xdot of the same shape as x.  Host-side helper kept for API parity (the reference
    exports it at module level); the rollouts never call it -- the kernels fuse it."""
    x, u = np.asarray(x, dtype=np.float64), np.asarray(u, dtype=np.float64)
    v = 0.5 * WHEEL_RADIUS * (u[0] + u[1])
    return np.array([v * np.cos(x[2]), v * np.sin(x[2]), (WHEEL_RADIUS / WHEEL_BASE) * (u[1] - u[0])])


def unicycle_dynamics(x, u):
    """Unicycle kinematics, control/src/mppi:33-36: u = [forward speed, turn rate].  Host-side helper, as dd_dynamics."""
    x, u = np.asarray(x, dtype=np.float64), np.asarray(u, dtype=np.float64)
    return np.array([np.cos(x[2]) * u[0], np.sin(x[2]) * u[0], u[1] + 0.0 * u[0]])


class _Model(object):
    """One of the two integrator / dynamics pairs ``control/src/mppi`` defines, usable exactly like the
    reference's module-level functions: as the ``model=`` constructor argument (:62) and as a callable
    ``model(x0 [3,N], u [2,N], dt) -> [3,N]`` (how perform_action uses it, :210-213).  A call runs the
    engine's own plant kernel (the reference's operation order, theta wrap included)."""

    _CACHE = 4   # engines kept per model token, least recently used first out

    def __init__(self, name, doc):
        self.name = name
        self.__doc__ = doc
        self._engines = {}

    def __repr__(self):
        return "<motion_planning_amd model %s>" % self.name

    def __call__(self, x0, u, dt):
        x0 = np.asarray(x0, dtype=np.float64)
        u = np.asarray(u, dtype=np.float64)
        single = x0.ndim == 1
        xs = x0.reshape(3, -1)
        us = u.reshape(2, -1)
        n = xs.shape[1]
        dt = float(dt)
        if not dt > 0.0:
            # the engine reads dt <= 0 as "1 / horizon" (control/src/mppi:67); a zero-length step is the wrapped identity
            if dt == 0.0:
                out = xs.copy()
                if self.name == "rk4":
                    out[2] = out[2] - (np.ceil((out[2] + np.pi) / (2.0 * np.pi)) - 1.0) * 2.0 * np.pi
                return out[:, 0] if single else out
            raise ValueError("dt must be >= 0 (the plant kernel integrates forward)")
        key = (n, dt)
        eng = self._engines.pop(key, None)
        if eng is None:
            while len(self._engines) >= self._CACHE:
                self._engines.pop(next(iter(self._engines))).close()
            eng = Engine(1, 6, n_agents=n, dt=dt, model=self.name, storage="f64")
        self._engines[key] = eng   # most recently used last
        eng.set_nominal_all(np.repeat(us.T[:, :, None], 6, axis=2))
        out = eng.plant_step(np.ascontiguousarray(xs.T)).T
        return out[:, 0] if single else out


rk4 = _Model("rk4", "rk4 over dd_dynamics, control/src/mppi:23-30, :39-54 (the node's model)")
euler = _Model("euler", "euler over unicycle_dynamics, control/src/mppi:33-36, :57-58")


_MODELS = {"rk4": _capi.MPPI_MODEL_DIFFDRIVE_RK4, "euler": _capi.MPPI_MODEL_UNICYCLE_EULER,
           _capi.MPPI_MODEL_DIFFDRIVE_RK4: _capi.MPPI_MODEL_DIFFDRIVE_RK4,
           _capi.MPPI_MODEL_UNICYCLE_EULER: _capi.MPPI_MODEL_UNICYCLE_EULER}

# which kernels a tick runs (include/mppi_hip.h MPPI_TICK_*): "lanes" = lane per sample (throughput),
# "scan" = lanes are timesteps, one kernel (small-K latency), "auto" = scan while A * K <= 14336 (T <= 64) / 5120 (T <= 256)
_TICK_PATHS = {"auto": _capi.MPPI_TICK_AUTO, "lanes": _capi.MPPI_TICK_LANES, "scan": _capi.MPPI_TICK_SCAN}
_STORAGE = {"f32": MPPI_STORE_F32, "f64": MPPI_STORE_F64, MPPI_STORE_F32: MPPI_STORE_F32,
            MPPI_STORE_F64: MPPI_STORE_F64}


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError("expected shape %s, got %s" % (tuple(shape), a.shape))
    return a


class Engine(object):
    """One libmppi_hip engine: A agents x K samples (this GPU's shard) x T horizon."""

    # which kernels a tick runs when the constructor is not told (include/mppi_hip.h MPPI_TICK_*)
    default_tick_path = "auto"
    # mppi_set_option switches every new engine gets (tests and A/B measurements set this; the product leaves it empty)
    default_options = {}

    def __init__(self, samples, horizon, n_agents=1, storage="f32", device=0, sample_offset=0,
                 dt=None, sigma=0.9, lam=0.001, model="rk4", tick_path=None, co_shards=None, options=None, agent_offset=0,
                 samples_total=0, **overrides):
        self._lib = _capi.load()
        cfg = _capi.default_config()
        cfg.n_agents, cfg.samples, cfg.horizon = int(n_agents), int(samples), int(horizon)
        cfg.storage = _STORAGE[storage]
        cfg.device = int(device)
        cfg.sample_offset = int(sample_offset)
        cfg.agent_offset = int(agent_offset)   # global index of local agent 0 (replica ranks: the noise streams are keyed by the global index)
        # the whole controller's samples per agent when this engine is one shard of it (0: it is the whole): size rules follow IT, so
        # every shard runs the unsplit controller's arithmetic (include/mppi_hip.h mppi_config.samples_total)
        cfg.samples_total = int(samples_total or 0)
        cfg.model = _MODELS[model]
        cfg.tick_path = _TICK_PATHS[tick_path if tick_path is not None else self.default_tick_path]
        # co-scheduled shards of the fused tick (include/mppi_hip.h): None = the engine's own rule, 1 = off, 2..8
        cfg.co_shards = 0 if co_shards is None else int(co_shards)
        cfg.dt = 0.0 if dt is None else float(dt)
        cfg.sigma, cfg.lambda_ = float(sigma), float(lam)
        for key, val in overrides.items():
            if key in ("q", "r", "p1"):
                arr = getattr(cfg, key)
                for i, v in enumerate(val):
                    arr[i] = float(v)
            elif key in ("u_max", "wheel_radius", "wheel_base", "floor_w"):
                setattr(cfg, key, float(val))
            else:
                raise TypeError("unknown engine option %r" % key)
        self._h = C.c_void_p()
        rc = self._lib.mppi_create(C.byref(cfg), C.byref(self._h))
        if rc:
            self._h = None
            _capi.check(rc, None)
        self.A, self.K, self.T = cfg.n_agents, cfg.samples, cfg.horizon
        self.dt = 1.0 / self.T if dt is None else float(dt)
        self.storage = "f64" if cfg.storage == MPPI_STORE_F64 else "f32"
        self.sigma, self.lam = cfg.sigma, cfg.lambda_
        # staging buffers of the blocking tick with their ctypes pointers made once (ndarray.ctypes.data_as costs 2 us a time:
        # four of them were a tenth of the node's 20 us call)
        self._io = [np.zeros((self.A, 3)), np.zeros((self.A, 3)), np.empty((self.A, 3)), np.empty((self.A, 2))]
        self._io_ptr = [_capi.dptr(a) for a in self._io]
        for key, val in dict(self.default_options, **(options or {})).items():
            self.set_option(key, val)

    def set_option(self, key, value):
        """A measurement / test switch of include/mppi_hip.h (mppi_set_option)."""
        self._ck(self._lib.mppi_set_option(self._h, key.encode(), int(value)))

    def get_option(self, key):
        v = C.c_int64()
        self._ck(self._lib.mppi_get_option(self._h, key.encode(), C.byref(v)))
        return int(v.value)

    # -- lifetime ------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.mppi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _ck(self, rc):
        _capi.check(rc, self._h)

    # -- configuration -------------------------------------------------------------------
    def set_stream(self, stream_ptr):
        self._ck(self._lib.mppi_set_stream(self._h, C.c_void_p(int(stream_ptr))))

    def get_stream(self):
        """The hipStream_t (as an integer) the engine enqueues on."""
        st = C.c_void_p()
        self._ck(self._lib.mppi_get_stream(self._h, C.byref(st)))
        return st.value or 0

    def set_sigma_lambda(self, sigma, lam):
        if sigma != self.sigma or lam != self.lam:
            self._ck(self._lib.mppi_set_sigma_lambda(self._h, float(sigma), float(lam)))
            self.sigma, self.lam = float(sigma), float(lam)
            self._sig_matrix = None

    def set_sig(self, sig, lam):
        """sig as get_path accepts it (control/src/mppi:88): a scalar sigma (= sigma * I) or any 2 x 2
        matrix -- the noise is drawn with sig[0,0] (:143-146), the stage cost uses the whole matrix (:184)."""
        sig = np.asarray(sig, dtype=np.float64)
        if sig.ndim == 0:
            return self.set_sigma_lambda(float(sig), lam)
        if sig.shape != (2, 2):
            raise ValueError("sig must be a scalar or a 2 x 2 matrix")
        if sig[0, 1] == 0.0 and sig[1, 0] == 0.0 and sig[0, 0] == sig[1, 1]:
            return self.set_sigma_lambda(float(sig[0, 0]), lam)
        m = _f64(sig, (2, 2))
        key = (m.tobytes(), float(lam))
        if getattr(self, "_sig_matrix", None) == key:
            return   # unchanged: the C call would settle the lazy state (a full re-rollout on the scan path) for nothing
        self._ck(self._lib.mppi_set_sig_matrix(self._h, _capi.dptr(m), float(lam)))
        self._sig_matrix = key
        self.sigma, self.lam = None, float(lam)   # not a scalar any more: the next set_sigma_lambda always applies

    def set_weights(self, q=None, r=None, p1=None):
        """Diagonals of Q [3], R [2], P1 [3] (control/src/mppi:69-73); None keeps the current one."""
        arrs = [None if v is None else _f64(v, (n,)) for v, n in ((q, 3), (r, 2), (p1, 3))]
        self._ck(self._lib.mppi_set_weights(self._h, *[_capi.dptr(a) for a in arrs]))

    def set_weight_matrices(self, Q=None, R=None, P1=None):
        """The full matrices Q [3][3], R [2][2], P1 [3][3] the reference multiplies (control/src/mppi:168, :181-184); None keeps
        the current one.  Off-diagonal terms select the general-cost rollout."""
        arrs = [None if v is None else _f64(v, (n, n)) for v, n in ((Q, 3), (R, 2), (P1, 3))]
        self._ck(self._lib.mppi_set_weight_matrices(self._h, *[_capi.dptr(a) for a in arrs]))

    def set_sync_timeout(self, milliseconds):
        self._ck(self._lib.mppi_set_sync_timeout(self._h, int(milliseconds)))

    def set_tick_counter(self, next_tick_id):
        self._ck(self._lib.mppi_set_tick_counter(self._h, int(next_tick_id)))

    def stream_wait_partials(self, other_stream_ptr):
        self._ck(self._lib.mppi_stream_wait_partials(self._h, C.c_void_p(int(other_stream_ptr))))

    def wait_for_stream(self, other_stream_ptr):
        self._ck(self._lib.mppi_wait_for_stream(self._h, C.c_void_p(int(other_stream_ptr))))

    def set_obstacle_grid(self, cells, resolution, origin, weight):
        """EXTENSION (not in the reference cost): occupancy grid in map::Grid's export format,
        cells [height][width] int8 in {0, 50, 100}; stage cost += weight * cell / 100.
        cells=None or weight=0 removes it."""
        if cells is None or weight == 0:
            self._ck(self._lib.mppi_set_obstacle_grid(self._h, None, 0, 0, 1.0, 0.0, 0.0, 0.0))
            return
        g = np.ascontiguousarray(cells, dtype=np.int8)
        if g.ndim != 2:
            raise ValueError("cells must be [height][width]")
        self._ck(self._lib.mppi_set_obstacle_grid(self._h, g.ctypes.data, g.shape[1], g.shape[0], float(resolution),
                                                  float(origin[0]), float(origin[1]), float(weight)))

    def reset(self, agent=-1):
        self._ck(self._lib.mppi_reset(self._h, int(agent)))

    def set_nominal(self, uvec, agent=0):
        u = _f64(uvec, (2, self.T))
        self._ck(self._lib.mppi_set_nominal(self._h, int(agent), _capi.dptr(u)))

    def set_shift_fill(self, fill, agent=0):
        """What the receding-horizon shift puts into the freed last column: uvec_init[:, 0] (control/src/mppi:101)."""
        f = _f64(fill, (2,))
        self._ck(self._lib.mppi_set_shift_fill(self._h, int(agent), _capi.dptr(f)))

    def set_nominal_all(self, uvecs):
        """uvecs [A][2][T]: the nominal controls of every agent."""
        u = _f64(uvecs, (self.A, 2, self.T))
        for a in range(self.A):
            self._ck(self._lib.mppi_set_nominal(self._h, a, _capi.dptr(u[a])))

    def get_nominal(self, agent=0):
        u = np.empty((2, self.T))
        self._ck(self._lib.mppi_get_nominal(self._h, int(agent), _capi.dptr(u)))
        return u

    # -- two-stage path (get_cost2go / update_action) --------------------------------------
    def upload_noise(self, eps):
        e = _f64(eps).reshape(self.A, self.T, 2, self.K)
        self._ck(self._lib.mppi_upload_noise(self._h, _capi.dptr(e)))

    def download_noise(self):
        e = np.empty((self.A, self.T, 2, self.K))
        self._ck(self._lib.mppi_download_noise(self._h, _capi.dptr(e)))
        return e

    def _sg(self, state, goal):
        s = None if state is None else _f64(state).reshape(self.A, 3)
        g = None if goal is None else _f64(goal).reshape(self.A, 3)
        return s, g

    def rollout(self, state, goal, noise="injected", seed=0, tick_id=0):
        s, g = self._sg(state, goal)
        mode = MPPI_NOISE_PHILOX if noise == "philox" else MPPI_NOISE_INJECTED
        self._ck(self._lib.mppi_rollout(self._h, _capi.dptr(s), _capi.dptr(g), mode, int(seed), int(tick_id)))

    def download_value(self):
        v = np.empty((self.A, self.T, self.K))
        self._ck(self._lib.mppi_download_value(self._h, _capi.dptr(v)))
        return v

    def upload_value(self, V):
        v = _f64(V).reshape(self.A, self.T, self.K)
        self._ck(self._lib.mppi_upload_value(self._h, _capi.dptr(v)))

    def update(self, want_output=True):
        u = np.empty((self.A, 2, self.T)) if want_output else None
        self._ck(self._lib.mppi_update(self._h, _capi.dptr(u)))
        return u

    def get_unfiltered(self):
        """[A][2][T]: the last update()'s controls before the filter (updated + clipped): what the reference's update_action leaves
        in its caller's uvec (control/src/mppi:196-199)."""
        out = np.empty((self.A, 2, self.T))
        self._ck(self._lib.mppi_get_unfiltered(self._h, _capi.dptr(out)))
        return out

    def plant_step(self, state=None):
        s, _ = self._sg(state, None)
        nxt = np.empty((self.A, 3))
        self._ck(self._lib.mppi_plant_step(self._h, _capi.dptr(s), _capi.dptr(nxt)))
        return nxt

    def shift(self):
        self._ck(self._lib.mppi_shift(self._h))

    # -- tick path -----------------------------------------------------------------------
    def tick_begin(self, state=None, goal=None, noise="philox", seed=0, tick_id=0):
        s, g = self._sg(state, goal)
        mode = MPPI_NOISE_PHILOX if noise == "philox" else MPPI_NOISE_INJECTED
        self._ck(self._lib.mppi_tick_begin(self._h, _capi.dptr(s), _capi.dptr(g), mode, int(seed), int(tick_id)))

    def partials(self):
        """(device pointer, bytes) of this shard's merged partials [A][T][8] float64."""
        p, n = C.c_void_p(), C.c_size_t()
        self._ck(self._lib.mppi_partials_ptr(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def tick_finish(self, gathered_ptr=None, n_shards=1):
        self._ck(self._lib.mppi_tick_finish(self._h, C.c_void_p(gathered_ptr) if gathered_ptr else None, int(n_shards)))

    # -- peer-to-peer exchange (K sharded over the GPUs of one node; include/mppi_hip.h mppi_p2p_*) ------------
    def p2p_create(self, n_ranks, rank):
        """Allocates this rank's mailbox; returns its HIP IPC handle (bytes)."""
        buf = C.create_string_buffer(_capi.IPC_HANDLE_BYTES)
        self._ck(self._lib.mppi_p2p_create(self._h, int(n_ranks), int(rank), buf))
        return bytes(buf.raw)

    def p2p_connect(self, handles=None, local_ptrs=None):
        """handles: one IPC handle (bytes) per rank, rank-major; local_ptrs: mailbox pointers of engines in this process."""
        blob = b"".join(handles) if handles is not None else None
        arr = None
        if local_ptrs is not None:
            arr = (C.c_void_p * len(local_ptrs))(*[C.c_void_p(p) if p else None for p in local_ptrs])
        self._ck(self._lib.mppi_p2p_connect(self._h, blob, arr))

    def p2p_rendezvous(self, path_prefix, n_ranks, rank, timeout_ms=60000):
        """p2p_create + connect for ranks in separate processes with only a file system in common (no process group)."""
        self._ck(self._lib.mppi_p2p_rendezvous(self._h, str(path_prefix).encode(), int(n_ranks), int(rank), int(timeout_ms)))

    def p2p_mailbox_ptr(self):
        p = C.c_void_p()
        self._ck(self._lib.mppi_p2p_mailbox_ptr(self._h, C.byref(p)))
        return p.value

    def p2p_selftest(self, rounds=4):
        self._ck(self._lib.mppi_p2p_selftest(self._h, int(rounds)))

    def p2p_destroy(self):
        self._ck(self._lib.mppi_p2p_destroy(self._h))

    def p2p_publish(self):
        self._ck(self._lib.mppi_p2p_publish(self._h))

    def tick_finish_p2p(self):
        self._ck(self._lib.mppi_tick_finish_p2p(self._h))

    def tick_exchange_p2p(self):
        self._ck(self._lib.mppi_tick_exchange_p2p(self._h))

    def get_outputs(self):
        nxt, ua = np.empty((self.A, 3)), np.empty((self.A, 2))
        self._ck(self._lib.mppi_get_outputs(self._h, _capi.dptr(nxt), _capi.dptr(ua)))
        return nxt, ua

    def tick(self, state=None, goal=None, noise="philox", seed=0, tick_id=0):
        """One blocking control tick: mppi_tick (= tick_begin + tick_finish + get_outputs in one call)."""
        sp, gp = self._stage(state, goal)
        mode = MPPI_NOISE_PHILOX if noise == "philox" else MPPI_NOISE_INJECTED
        rc = self._lib.mppi_tick(self._h, sp, gp, mode, int(seed), int(tick_id), self._io_ptr[2], self._io_ptr[3])
        if rc:
            _capi.check(rc, self._h)
        return self._io[2].copy(), self._io[3].copy()

    def _stage(self, state, goal):
        """state / goal copied into the engine object's own staging arrays (the library reads them during the call)."""
        sp = gp = None
        if state is not None:
            self._io[0][...] = np.asarray(state, dtype=np.float64).reshape(self.A, 3)
            sp = self._io_ptr[0]
        if goal is not None:
            self._io[1][...] = np.asarray(goal, dtype=np.float64).reshape(self.A, 3)
            gp = self._io_ptr[1]
        return sp, gp

    def tick_async(self, state=None, goal=None, noise="philox", seed=0, tick_id=0):
        """The fused tick without waiting for its outputs (mppi_tick with NULL outputs): everything is enqueued, nothing
        blocks; get_outputs() later.  Unlike tick_begin + tick_finish it has no exchange point, so the engine may take
        its shortcuts (zero-copy inputs, no merge launch for a handful of tuples)."""
        sp, gp = self._stage(state, goal)
        mode = MPPI_NOISE_PHILOX if noise == "philox" else MPPI_NOISE_INJECTED
        rc = self._lib.mppi_tick(self._h, sp, gp, mode, int(seed), int(tick_id), None, None)
        if rc:
            _capi.check(rc, self._h)

    def tick_graph(self, seed=0):
        self._ck(self._lib.mppi_tick_graph(self._h, int(seed)))

    def synchronize(self):
        self._ck(self._lib.mppi_synchronize(self._h))

    # -- instrumentation -----------------------------------------------------------------
    def kernel_timing(self, kernels=(), period=1):
        self._ck(self._lib.mppi_kernel_timing_period(self._h, int(period)))
        mask = 0
        for k in kernels:
            mask |= 1 << _capi.KERNELS.index(k)
        self._ck(self._lib.mppi_kernel_timing(self._h, mask))

    def kernel_times(self):
        ms = (C.c_double * len(_capi.KERNELS))()
        n = (C.c_int64 * len(_capi.KERNELS))()
        self._ck(self._lib.mppi_kernel_times(self._h, ms, n))
        return {k: (ms[i], n[i]) for i, k in enumerate(_capi.KERNELS)}

    def shader_clock_mhz(self):
        """Shader clock the last lane-per-sample rollout launch ran at (a probe wave inside that launch), MHz."""
        mhz = C.c_double()
        self._ck(self._lib.mppi_shader_clock(self._h, C.byref(mhz)))
        return mhz.value

    def probe_timeline(self):
        """(stamps [30], total): shader cycles of the last rollout launch's probe wave at its marks (diagnostic builds of the
        library only -- make PROBE=1; the product build returns zeros)."""
        marks, total = (C.c_uint64 * _capi.PROBE_MARKS)(), C.c_uint64()
        self._ck(self._lib.mppi_probe_timeline(self._h, marks, C.byref(total)))
        return [int(v) for v in marks], int(total.value)

    def info(self):
        b, r, u = C.c_size_t(), C.c_int32(), C.c_int32()
        self._ck(self._lib.mppi_engine_info(self._h, C.byref(b), C.byref(r), C.byref(u)))
        n, per = C.c_int32(), (C.c_int32 * 8)()
        self._ck(self._lib.mppi_co_info(self._h, C.byref(n), per))
        kind = C.c_int32()
        self._ck(self._lib.mppi_rollout_kernel(self._h, C.byref(kind)))
        return {"hbm_bytes": b.value, "rollout_blocks": r.value, "update_blocks": u.value,
                "tick_kernels": "scan" if u.value == 0 else "lanes",
                "co_shards": n.value, "co_samples": [per[g] for g in range(n.value)],
                "co_note": (self._lib.mppi_co_note(self._h) or b"").decode(),
                # which kernel the last rollout launch was (include/mppi_hip.h MPPI_ROLLOUT_*)
                "rollout_kernel": ("none", "fp64", "mixed", "scan", "fused")[kind.value]}


class _NominalView(np.ndarray):
    """Host copy of the device-resident nominal controls whose assignments write through."""

    def __new__(cls, arr, engine):
        obj = np.asarray(arr, dtype=np.float64).view(cls)
        obj._engine = engine
        obj._root = obj
        return obj

    def __array_finalize__(self, obj):
        self._engine = getattr(obj, "_engine", None)
        self._root = getattr(obj, "_root", None)

    def __setitem__(self, key, value):
        super().__setitem__(key, value)
        root = self._root
        if root is not None and self._engine is not None and np.may_share_memory(self, root):
            self._engine.set_nominal(np.asarray(root))


def savgol_matrix(horizon):
    """The operator S with savgol_filter(u, T-1, 3, axis=1) == u @ S (control/src/mppi:202)."""
    S = np.empty((horizon, horizon))
    rc = _capi.load().mppi_savgol_matrix(int(horizon), _capi.dptr(S))
    if rc:
        raise ValueError("horizon %d: the Savitzky-Golay window horizon - 1 must be > 3" % horizon)
    return S


class MPPI(object):
    """Drop-in for the reference ``MPPI`` (control/src/mppi:61-213), running on the GPU.

    Extra keyword arguments (not in the reference):
      rng      "numpy"  -- draw noise exactly like the reference (np.random.normal per
                           timestep from numpy's global RNG, :143-146) and inject it: same
                           seed => same trajectory as the reference;
               "philox" -- device Philox4x32-10 keyed by (seed, tick, sample): nothing
                           crosses PCIe, the production mode.
      storage  "f32" (default) | "f64"  HBM storage of eps / V (arithmetic is fp64 either way).
    update_action mutates its arguments like the reference does (value_fcn rows minus their minimum, uvec plus the weighted noise,
    clipped: control/src/mppi:189, :196-199) when they are writable numpy arrays.
    """

    def __init__(self, model=rk4, horizon=100, samples=10, thresh=0.05, rng="numpy", seed=0,
                 storage="f32", device=0, tick_path=None):
        if isinstance(model, _Model):
            model_id = model.name
        elif model in ("rk4", "euler"):
            model_id = model
        else:
            raise NotImplementedError("model must be rk4 (diff drive, the node's default) or euler (unicycle): "
                                      "the two integrators control/src/mppi defines")
        if rng not in ("numpy", "philox"):
            raise ValueError("rng must be 'numpy' or 'philox'")
        self.horizon = int(horizon)
        self.samples = int(samples)
        self.uvec_init = np.zeros((2, self.horizon))
        self.model = model
        self.dt = 1.0 / float(horizon)
        self._eng = None
        self._weights_sent = None
        self._weights_raw = None        # the raw bytes of Q, R, P1 when they were last validated (the per-tick fast path)
        self._fill_sent = np.zeros(2)   # the engine starts with a zero fill
        self._path_buf = self._uvec_buf = None
        self.Q = np.array([[1e3, 0.0, 0.0], [0.0, 1e3, 0.0], [0.0, 0.0, 0.0]])
        self.R = np.array([[1.0, 0.0], [0.0, 1.0]])
        self.P1 = np.array([[1e3, 0.0, 0.0], [0.0, 1e3, 0.0], [0.0, 0.0, 1e3]])
        self.thresh = thresh
        self.start = np.array([0.0, 0.0, 0.0])
        self.goal = np.array([0.0, 0.0, 0.0])
        self.rng = rng
        self.seed = int(seed)
        self._tick = 0
        self._eng = Engine(self.samples, self.horizon, 1, storage=storage, device=device, model=model_id,
                           tick_path=tick_path)
        self.initialize()

    # Q, R, P1 (control/src/mppi:69-73) are plain instance attributes in the reference, read on every call (:168, :183):
    # ``m.Q = ...`` or ``m.Q[0, 0] = ...`` must change the cost of the next rollout here too.  The attributes stay plain
    # arrays; every call that rolls out hands their current values to the engine first when they differ from what it has:
    # diagonal matrices by their diagonals (mppi_set_weights: the fast kernels), anything else whole (mppi_set_weight_matrices:
    # the general-cost rollout carries the symmetric parts, which is all a quadratic form sees).
    def _sync_weights(self):
        # fast path (every tick of the node): the three attributes are still float64 arrays holding the bytes last validated and sent
        raw = self._weights_raw
        if raw is not None:
            q, r, p1 = self.Q, self.R, self.P1
            if (type(q) is np.ndarray and type(r) is np.ndarray and type(p1) is np.ndarray and q.dtype == np.float64 and
                    r.dtype == np.float64 and p1.dtype == np.float64 and q.tobytes() == raw[0] and r.tobytes() == raw[1] and
                    p1.tobytes() == raw[2]):
                return
        mats, diagonal = [], True
        for name, n in (("Q", 3), ("R", 2), ("P1", 3)):
            m = np.array(getattr(self, name), dtype=np.float64)
            if m.shape != (n, n):
                raise ValueError("%s must be %d x %d (control/src/mppi:69-73)" % (name, n, n))
            diagonal = diagonal and not np.count_nonzero(m - np.diag(np.diag(m)))
            mats.append(m)
        key = tuple(map(bytes, (v.tobytes() for v in mats)))
        if key != self._weights_sent:
            if diagonal:
                self._eng.set_weights(*[np.diag(m).copy() for m in mats])
            else:
                self._eng.set_weight_matrices(*mats)
            self._weights_sent = key
        self._weights_raw = tuple(np.asarray(getattr(self, name), dtype=np.float64).tobytes() for name in ("Q", "R", "P1"))

    # uvec_init[:, 0] is what every receding-horizon shift appends, read live on each get_path (control/src/mppi:101): an
    # assignment to m.uvec_init (or into it) between calls must reach the engine without an initialize()
    def _sync_fill(self):
        init = self.uvec_init
        if (type(init) is np.ndarray and init.dtype == np.float64 and init.ndim == 2 and init.shape[0] == 2 and init.shape[1] >= 1
                and self._fill_sent is not None and init[0, 0] == self._fill_sent[0] and init[1, 0] == self._fill_sent[1]):
            return   # (every tick of the node)
        init = np.asarray(self.uvec_init, dtype=np.float64)
        if init.ndim != 2 or init.shape[0] != 2 or init.shape[1] < 1:
            raise ValueError("uvec_init must be [2, horizon] (control/src/mppi:65)")
        fill = init[:, 0].copy()
        if self._fill_sent is None or np.any(fill != self._fill_sent):
            self._eng.set_shift_fill(fill)
            self._fill_sent = fill

    def _load_uvec_init(self, init):
        if np.any(init):
            self._eng.set_nominal(init)
        else:
            self._eng.reset()                 # the reference's default: zeros (one asynchronous memset)
        fill = init[:, 0].copy()
        if self._fill_sent is None or np.any(fill != self._fill_sent):
            self._eng.set_shift_fill(fill)
            self._fill_sent = fill

    # control/src/mppi:79-83
    def initialize(self):
        self.fin_time = [0]
        init = np.asarray(self.uvec_init, dtype=np.float64)
        if init.shape != (2, self.horizon):
            raise ValueError("uvec_init must be [2, horizon] (control/src/mppi:65)")
        self._load_uvec_init(init)           # latest_uvec = uvec_init (:81); every later shift appends uvec_init[:, 0] (:101)
        self.uvec = np.array([init[:, 0]])
        self.path = np.array([self.start])

    # path [n][3] and uvec [n][2] grow by one row per get_path (control/src/mppi:97-98: np.concatenate, O(n) per tick).  They are
    # views of buffers that double when full; assigning to the attributes works as it does on the reference's plain arrays.
    @staticmethod
    def _grown(buf, n, row):
        if buf is None or n >= buf.shape[0]:
            new = np.empty((max(64, 2 * n), len(row)))
            if buf is not None:
                new[:n] = buf[:n]
            buf = new
        buf[n] = row
        return buf

    @property
    def path(self):
        return self._path_buf[:self._path_n]

    @path.setter
    def path(self, value):
        value = np.array(value, dtype=np.float64, ndmin=2)
        self._path_buf, self._path_n = value, value.shape[0]

    @property
    def uvec(self):
        return self._uvec_buf[:self._uvec_n]

    @uvec.setter
    def uvec(self, value):
        value = np.array(value, dtype=np.float64, ndmin=2)
        self._uvec_buf, self._uvec_n = value, value.shape[0]

    @property
    def latest_uvec(self):
        """The nominal control sequence [2, T] (control/src/mppi:81).  It lives on the device; what comes
        back is an array whose item / slice assignments write through (``m.latest_uvec[:, 0] = 0`` works
        like it does on the reference's attribute)."""
        return _NominalView(self._eng.get_nominal(), self._eng)

    @latest_uvec.setter
    def latest_uvec(self, u):
        self._eng.set_nominal(u)

    def _set_sig(self, sig, lam):
        """Hands sig / lam to the engine; returns the std-dev the noise is drawn with = sig[0,0] (:145)."""
        sig = np.asarray(sig, dtype=np.float64)
        if sig.ndim not in (0, 2) or (sig.ndim == 2 and sig.shape != (2, 2)):
            raise ValueError("sig must be a scalar or the 2 x 2 matrix get_path takes (control/src/mppi:88)")
        self._eng.set_sig(sig, lam)
        return float(sig) if sig.ndim == 0 else float(sig[0, 0])

    def _draw(self, sigma):
        # one legacy-RNG call per timestep, exactly the reference's consumption (:143-146)
        return np.stack([np.random.normal(0, sigma, size=(2, self.samples)) for _ in range(self.horizon)])

    # control/src/mppi:85-102
    def get_path(self, state, goal, sig=np.array([[.9, 0.0], [0.0, .9]]), lam=.001):
        self._sync_weights()
        self._sync_fill()
        sigma = self._set_sig(sig, lam)
        if self.rng == "numpy":
            self._eng.upload_noise(self._draw(sigma))
            nxt, ua = self._eng.tick(state, goal, noise="injected")
        else:
            nxt, ua = self._eng.tick(state, goal, noise="philox", seed=self.seed, tick_id=self._tick)
        self._tick += 1
        state = nxt[0]
        self._path_buf = self._grown(self._path_buf, self._path_n, state)       # :97
        self._path_n += 1
        self._uvec_buf = self._grown(self._uvec_buf, self._uvec_n, ua[0])       # :98
        self._uvec_n += 1
        self.fin_time.append(self.fin_time[-1] + self.dt)
        return state

    # control/src/mppi:104-125
    def solve_path(self, start, goal, sig=np.array([[1.0, 0.0], [0.0, 1.0]]), lam=.01, max_iters=100000):
        self.start = np.asarray(start, dtype=np.float64)
        self.goal = np.asarray(goal, dtype=np.float64)
        state = self.start
        self.path = np.array([state])
        init = np.asarray(self.uvec_init, dtype=np.float64)
        self._load_uvec_init(init)            # latest_uvec = uvec_init (:114)
        i = 0
        while np.linalg.norm(state[:2] - self.goal[:2]) > self.thresh and i < max_iters:
            i += 1
            state = self.get_path(state, self.goal, sig, lam)
        return state, i

    # control/src/mppi:127-178
    def get_cost2go(self, state, uvec, goal, lam, sig):
        self._sync_weights()
        sigma = self._set_sig(sig, lam)
        self._eng.set_nominal(uvec)
        if self.rng == "numpy":
            self._eng.upload_noise(self._draw(sigma))
            self._eng.rollout(state, goal, noise="injected")
        else:
            self._eng.rollout(state, goal, noise="philox", seed=self.seed, tick_id=self._tick)
            self._tick += 1
        value_fcn = self._eng.download_value()[0]
        eps = self._eng.download_noise()[0]
        return value_fcn, [eps[t] for t in range(self.horizon)]

    # control/src/mppi:180-184 -- scalar form kept for API parity; the kernels fuse it
    def get_cost(self, state, desired_state, u, lam, sig, eps):
        d = np.asarray(state, dtype=np.float64) - np.asarray(desired_state, dtype=np.float64)
        u = np.asarray(u, dtype=np.float64)
        return 0.5 * (d.dot(self.Q).dot(d) + u.dot(self.R).dot(u)) + lam * u.dot(np.asarray(sig)).dot(eps)

    # control/src/mppi:186-208
    def update_action(self, uvec, eps, value_fcn, sig, lam):
        self._set_sig(sig, lam)
        self._eng.set_nominal(uvec)
        self._eng.upload_noise(np.asarray(eps, dtype=np.float64))
        self._eng.upload_value(value_fcn)
        out = self._eng.update()[0]
        # the reference's side effects on its arguments (control/src/mppi:189, :196-199): every row of value_fcn has its minimum taken
        # out IN PLACE, uvec receives the weighted noise and the first clip IN PLACE; the filtered sequence is returned as a new array
        if isinstance(value_fcn, np.ndarray) and value_fcn.flags.writeable:
            value_fcn -= value_fcn.min(axis=1, keepdims=True)
        if isinstance(uvec, np.ndarray) and uvec.flags.writeable:
            uvec[...] = self._eng.get_unfiltered()[0]
        return out

    # control/src/mppi:210-213
    def perform_action(self, state, uvec):
        keep = self._eng.get_nominal()  # the reference's perform_action has no side effects
        self._eng.set_nominal(uvec)
        nxt = self._eng.plant_step(state)[0]
        self._eng.set_nominal(keep)
        return nxt
