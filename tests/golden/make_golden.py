#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the REAL reference.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

The reference MPPI controller is the extension-less rospy script
``/root/reference/control/src/mppi``.  ROS is not installed here, so the
ROS modules it imports are replaced by empty stand-in *modules* (they are only
touched by ``Controller``/``main``, which we drive by hand below); the numeric
code (``dd_dynamics``, ``rk4``, ``MPPI``) runs unmodified.  Nothing of the
reference is copied: the fixtures hold inputs and expected outputs only.

Noise convention: the reference draws
``np.random.normal(0, sig[0,0], size=(2, K))`` once per timestep from numpy's
legacy global MT19937.  For an even draw count per call that stream equals
``np.random.RandomState(seed).normal(0, sigma, (n_ticks, T, 2, K))`` -- asserted
below -- so fixtures store only the seed; tests regenerate the noise.
"""
import importlib.machinery
import importlib.util
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/control/src/mppi"


# --------------------------------------------------------------------------
# import recipe
# --------------------------------------------------------------------------
class _Rec(object):
    """Attribute bag standing in for a ROS message."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


class _Pub(object):
    def __init__(self, *a, **k):
        self.sent = []

    def publish(self, msg):
        self.sent.append((msg.linear.x, msg.angular.z))


def _twist():
    return _Rec(linear=_Rec(x=0.0, y=0.0, z=0.0), angular=_Rec(x=0.0, y=0.0, z=0.0))


def _euler_from_quaternion(q):
    # yaw-only stand-in for tf.transformations.euler_from_quaternion (external
    # ROS library); the harness below only ever feeds pure-yaw quaternions.
    x, y, z, w = q
    yaw = np.arctan2(2.0 * (w * z + x * y), 1.0 - 2.0 * (y * y + z * z))
    return (0.0, 0.0, yaw)


def load_reference(params):
    import matplotlib
    matplotlib.use("Agg")
    rospy = types.ModuleType("rospy")
    rospy.Subscriber = lambda *a, **k: None
    rospy.Publisher = _Pub
    rospy.get_param = lambda name: params[name]
    rospy.loginfo = lambda *a, **k: None
    rospy.ROSInterruptException = type("ROSInterruptException", (Exception,), {})
    tf = types.ModuleType("tf")
    tft = types.ModuleType("tf.transformations")
    tft.euler_from_quaternion = _euler_from_quaternion
    tf.transformations = tft
    nav = types.ModuleType("nav_msgs")
    navm = types.ModuleType("nav_msgs.msg")
    navm.Odometry = _Rec
    nav.msg = navm
    geo = types.ModuleType("geometry_msgs")
    geom = types.ModuleType("geometry_msgs.msg")
    geom.Twist = _twist
    geom.Quaternion = _Rec
    geom.Vector3 = _Rec
    geo.msg = geom
    for name, mod in [("rospy", rospy), ("tf", tf), ("tf.transformations", tft),
                      ("nav_msgs", nav), ("nav_msgs.msg", navm),
                      ("geometry_msgs", geo), ("geometry_msgs.msg", geom)]:
        sys.modules[name] = mod
    loader = importlib.machinery.SourceFileLoader("ref_mppi", REF)
    spec = importlib.util.spec_from_loader("ref_mppi", loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    return mod


def nominal_warm(T):
    return np.array([np.linspace(-2.0, 1.0, T), np.linspace(1.5, -1.0, T)])


def nominal_hot(T):
    # drives the sampled controls into the +-6.35492 clip
    return np.array([np.linspace(5.5, 6.3, T), np.linspace(-6.3, -5.0, T)])


SIG = 0.9
LAM = 0.001


def main():
    params = {"waypoints": []}
    ref = load_reference(params)
    out = {}
    kat = {}

    # ---------------------------------------------------------------- A: KATs
    np.random.seed(0)
    m = ref.MPPI()
    s1 = m.get_path(np.array([0.0, 0.0, 0.0]), np.array([0.0, -1.0, 0.0]))
    u1 = m.uvec[-1].copy()
    kat["default_tick1_state"] = s1.tolist()
    kat["default_tick1_u"] = u1.tolist()
    kat["default_tick1_sum_latest_uvec"] = float(m.latest_uvec.sum())
    s2 = m.get_path(s1, np.array([0.0, -1.0, 0.0]))
    kat["default_tick2_state"] = s2.tolist()
    kat["default_tick2_u"] = m.uvec[-1].tolist()
    np.random.seed(0)
    m = ref.MPPI(horizon=50, samples=64)
    s = m.get_path(np.array([0.0, 0.0, 0.0]), np.array([1.0, 0.0, 0.0]))
    kat["h50k64_tick1_state"] = s.tolist()
    kat["h50k64_tick1_u"] = m.uvec[-1].tolist()
    kat["rk4_wrap"] = ref.rk4(np.array([[0.0], [0.0], [3.1]]),
                              np.array([[-6.35492], [6.35492]]), 0.02)[:, 0].tolist()
    kat["rk4_plain"] = ref.rk4(np.array([[0.3], [-0.2], [0.7]]),
                               np.array([[1.25], [-0.5]]), 0.01)[:, 0].tolist()
    kat["dd_dynamics"] = ref.dd_dynamics(np.array([[0.3], [-0.2], [0.7]]),
                                         np.array([[1.25], [-0.5]]))[:, 0].tolist()
    ctl = ref.Controller()
    kat["wheelsToTwist_1_2"] = list(ctl.wheelsToTwist([1.0, 2.0]))
    kat["constants"] = {"WHEEL_VEL_MAX": ref.WHEEL_VEL_MAX, "WHEEL_RADIUS": ref.WHEEL_RADIUS,
                        "WHEEL_BASE": ref.WHEEL_BASE}

    # ------------------------------------------- B: get_cost2go / update_action
    cases = [
        # name, K, T, seed, state, goal, nominal
        ("c2g_zero", 16, 50, 0, [0.0, 0.0, 0.0], [0.0, -1.0, 0.0], "zero"),
        ("c2g_warm", 32, 50, 1, [0.1, -0.2, 0.3], [1.0, 0.0, 0.0], "warm"),
        ("c2g_wrap", 8, 100, 2, [0.0, 0.0, 3.1], [-1.0, 0.5, -3.0], "warm"),
        ("c2g_clip", 64, 20, 3, [-0.5, 0.25, -1.0], [0.5, 0.5, 1.0], "hot"),
        ("c2g_tiny", 2, 6, 4, [0.0, 0.0, 0.0], [0.2, 0.1, 0.0], "warm"),
    ]
    for name, K, T, seed, state, goal, nom in cases:
        mp = ref.MPPI(horizon=T, samples=K)
        u0 = {"zero": np.zeros((2, T)), "warm": nominal_warm(T), "hot": nominal_hot(T)}[nom]
        sig = np.array([[SIG, 0.0], [0.0, SIG]])
        np.random.seed(seed)
        V, eps = mp.get_cost2go(np.array(state), u0.copy(), np.array(goal), LAM, sig)
        eps = np.array(eps)
        chk = np.random.RandomState(seed).normal(0.0, SIG, (T, 2, K))
        assert np.array_equal(eps, chk), name
        V_in = V.copy()
        u_arg, V_arg = u0.copy(), V.copy()
        unew = mp.update_action(u_arg, list(eps), V_arg, sig, LAM)
        out[name + "_V"] = V_in
        out[name + "_unew"] = unew
        # (round 6) update_action's side effects on its ARGUMENTS (:189, :196-199): every row of value_fcn minus its minimum, uvec plus
        # the weighted noise, clipped -- both in place; the filtered sequence it returns is a new array
        out[name + "_V_inplace"] = V_arg
        out[name + "_u_inplace"] = u_arg
        out[name + "_meta"] = np.array([K, T, seed], dtype=np.int64)
        out[name + "_state"] = np.array(state)
        out[name + "_goal"] = np.array(goal)
        out[name + "_u0"] = u0

    # ---------------------------------------------- C: closed-loop tick sequences
    seqs = [
        ("seq_park", 32, 50, 5, 12, [0.0, 0.0, 0.0], [0.0, -1.0, 0.0]),
        ("seq_wp", 16, 100, 6, 8, [0.0, 0.0, 0.0], [1.0, 0.0, 0.0]),
    ]
    for name, K, T, seed, nt, state, goal in seqs:
        mp = ref.MPPI(horizon=T, samples=K)
        np.random.seed(seed)
        st = np.array(state)
        states, us, lat = [], [], []
        for _ in range(nt):
            st = mp.get_path(st, np.array(goal))
            states.append(st.copy())
            us.append(mp.uvec[-1].copy())
            lat.append(mp.latest_uvec.copy())
        # stream identity across ticks
        allnoise = np.random.RandomState(seed).normal(0.0, SIG, (nt, T, 2, K))
        nxt = np.random.normal(0.0, SIG, size=(2, K))
        rs = np.random.RandomState(seed)
        rs.normal(0.0, SIG, (nt, T, 2, K))
        assert np.array_equal(nxt, rs.normal(0.0, SIG, (2, K)))
        del allnoise
        out[name + "_states"] = np.array(states)
        out[name + "_u"] = np.array(us)
        out[name + "_latest_uvec"] = np.array(lat)
        out[name + "_meta"] = np.array([K, T, seed, nt], dtype=np.int64)
        out[name + "_state0"] = np.array(state)
        out[name + "_goal"] = np.array(goal)

    # ------------------------------------------------- D: Savitzky-Golay operator
    from scipy.signal import savgol_filter
    for T in (6, 10, 20, 50, 100, 200):
        # row i = response to the unit impulse e_i  =>  u_f = u @ S
        out["savgol_S_%d" % T] = savgol_filter(np.eye(T), T - 1, 3, axis=1)
    rs = np.random.RandomState(7)
    u = rs.uniform(-6.0, 6.0, (2, 50))
    out["savgol_in_50"] = u
    out["savgol_out_50"] = savgol_filter(u, 49, 3, axis=1)

    # ----------------------------------- E: BASELINE config 1 (K=1000, T=50) summary
    K, T, seed = 1000, 50, 0
    for nom in ("zero", "warm"):
        mp = ref.MPPI(horizon=T, samples=K)
        u0 = np.zeros((2, T)) if nom == "zero" else nominal_warm(T)
        mp.latest_uvec = u0.copy()
        np.random.seed(seed)
        sig = np.array([[SIG, 0.0], [0.0, SIG]])
        V, eps = mp.get_cost2go(np.array([0.0, 0.0, 0.0]), u0.copy(),
                                np.array([0.0, -1.0, 0.0]), LAM, sig)
        out["c1_%s_Vsum_t" % nom] = V.sum(axis=1)
        out["c1_%s_Vmin_t" % nom] = V.min(axis=1)
        out["c1_%s_Vcols" % nom] = V[:, ::100].copy()
        unew = mp.update_action(u0.copy(), eps, V.copy(), sig, LAM)
        out["c1_%s_unew" % nom] = unew
        # full tick
        mp = ref.MPPI(horizon=T, samples=K)
        mp.latest_uvec = u0.copy()
        np.random.seed(seed)
        nxt = mp.get_path(np.array([0.0, 0.0, 0.0]), np.array([0.0, -1.0, 0.0]))
        out["c1_%s_next_state" % nom] = nxt
        out["c1_%s_u_applied" % nom] = mp.uvec[-1].copy()

    # --------------------------------- F: Controller state machine (model in the loop)
    def run_controller(waypoints, seed, n_cb, K=None, T=None, thresh=None):
        params["waypoints"] = waypoints
        np.random.seed(seed)
        c = ref.Controller()
        if K is not None:
            c.mppi = ref.MPPI(horizon=T, samples=K, thresh=thresh)
        plant = np.array([0.0, 0.0, 0.0])
        rows = []
        for _ in range(n_cb):
            q = (0.0, 0.0, np.sin(plant[2] / 2.0), np.cos(plant[2] / 2.0))
            odom = _Rec(pose=_Rec(pose=_Rec(
                position=_Rec(x=plant[0], y=plant[1], z=0.0),
                orientation=_Rec(x=q[0], y=q[1], z=q[2], w=q[3]))))
            n0 = len(c.tw_pub.sent)
            c.pos_cb(odom)
            assert len(c.tw_pub.sent) == n0 + 1
            vx, wz = c.tw_pub.sent[-1]
            seen = c.mppi.start.copy()
            u = np.array([0.0, 0.0]) if c.done else c.mppi.uvec[-1, :].copy()
            rows.append(np.concatenate([seen, c.mppi.goal, u, [vx, wz, float(c.idx),
                                                                  float(c.done), float(c.init)]]))
            # plant: the reference's own integrator driven by the applied wheel speeds
            plant = ref.rk4(plant.reshape(3, 1), u.reshape(2, 1), c.mppi.dt)[:, 0]
        return np.array(rows)

    out["ctl_park"] = run_controller([], 11, 40, K=16, T=20, thresh=0.05)
    out["ctl_park_meta"] = np.array([16, 20, 11, 40], dtype=np.int64)
    # pentagon with a generous threshold so several waypoint switches happen quickly
    out["ctl_wp"] = run_controller([[1, 0], [2, 1], [1, 2], [0, 2], [0, 0]], 12, 60,
                                   K=16, T=20, thresh=0.97)
    out["ctl_wp_meta"] = np.array([16, 20, 12, 60], dtype=np.int64)

    # ---------------- G: the alternative model of the ctor argument: euler + unicycle (:33-36, :57-58)
    K, T, seed = 32, 50, 8
    mp = ref.MPPI(model=ref.euler, horizon=T, samples=K)
    u0 = nominal_warm(T)
    sig = np.array([[SIG, 0.0], [0.0, SIG]])
    np.random.seed(seed)
    V, eps = mp.get_cost2go(np.array([0.1, -0.2, 0.3]), u0.copy(), np.array([1.0, 0.5, 0.2]), LAM, sig)
    assert np.array_equal(np.array(eps), np.random.RandomState(seed).normal(0.0, SIG, (T, 2, K)))
    out["euler_c2g_V"] = V.copy()
    out["euler_c2g_unew"] = mp.update_action(u0.copy(), eps, V.copy(), sig, LAM)
    out["euler_c2g_meta"] = np.array([K, T, seed], dtype=np.int64)
    out["euler_c2g_state"] = np.array([0.1, -0.2, 0.3])
    out["euler_c2g_goal"] = np.array([1.0, 0.5, 0.2])
    out["euler_c2g_u0"] = u0
    K, T, seed, nt = 16, 20, 9, 6
    mp = ref.MPPI(model=ref.euler, horizon=T, samples=K)
    np.random.seed(seed)
    st = np.array([0.0, 0.0, 2.5])
    states, us = [], []
    for _ in range(nt):
        st = mp.get_path(st, np.array([-0.5, 0.4, 3.0]))
        states.append(st.copy())
        us.append(mp.uvec[-1].copy())
    out["euler_seq_states"] = np.array(states)
    out["euler_seq_u"] = np.array(us)
    out["euler_seq_meta"] = np.array([K, T, seed, nt], dtype=np.int64)
    kat["euler_step"] = ref.euler(np.array([[0.3], [-0.2], [3.1]]), np.array([[1.25], [-0.5]]), 0.25)[:, 0].tolist()

    # ---------------- H: the reference's own cost formulas with OTHER weights: Q, R, P1 are instance
    # attributes (:69-73); overwriting them after construction runs get_cost / the terminal cost
    # (:165-173, :180-184) unmodified on an anisotropic Q with a heading weight
    K, T, seed = 48, 50, 13
    mp = ref.MPPI(horizon=T, samples=K)
    mp.Q = np.diag([350.0, 900.0, 15.0])
    mp.R = np.diag([0.5, 2.0])
    mp.P1 = np.diag([800.0, 1200.0, 300.0])
    u0 = nominal_warm(T)
    state, goal = np.array([0.1, -0.05, 2.9]), np.array([0.4, -1.0, -2.8])
    np.random.seed(seed)
    V, eps = mp.get_cost2go(state, u0.copy(), goal, LAM, sig)
    assert np.array_equal(np.array(eps), np.random.RandomState(seed).normal(0.0, SIG, (T, 2, K)))
    out["wts_c2g_V"] = V.copy()
    out["wts_c2g_unew"] = mp.update_action(u0.copy(), eps, V.copy(), sig, LAM)
    out["wts_c2g_meta"] = np.array([K, T, seed], dtype=np.int64)
    out["wts_c2g_state"], out["wts_c2g_goal"], out["wts_c2g_u0"] = state, goal, u0
    out["wts_q"], out["wts_r"], out["wts_p1"] = np.diag(mp.Q).copy(), np.diag(mp.R).copy(), np.diag(mp.P1).copy()
    nt = 5
    mp.initialize()
    np.random.seed(seed + 1)
    st, states, us = state.copy(), [], []
    for _ in range(nt):
        st = mp.get_path(st, goal)
        states.append(st.copy())
        us.append(mp.uvec[-1].copy())
    out["wts_seq_states"], out["wts_seq_u"] = np.array(states), np.array(us)
    out["wts_seq_meta"] = np.array([K, T, seed + 1, nt], dtype=np.int64)

    # ---------------- I: another temperature and noise level (the sig / lam arguments of get_path, :88-89)
    K, T, seed, nt = 40, 50, 21, 5
    sig2, lam2 = 0.4, 0.02
    mp = ref.MPPI(horizon=T, samples=K)
    np.random.seed(seed)
    st, goal = np.array([0.2, 0.1, -0.4]), np.array([-0.3, -0.8, 1.0])
    states, us = [], []
    for _ in range(nt):
        st = mp.get_path(st, goal, sig=np.array([[sig2, 0.0], [0.0, sig2]]), lam=lam2)
        states.append(st.copy())
        us.append(mp.uvec[-1].copy())
    out["lamsig_seq_states"], out["lamsig_seq_u"] = np.array(states), np.array(us)
    out["lamsig_seq_latest_uvec"] = mp.latest_uvec.copy()
    out["lamsig_seq_meta"] = np.array([K, T, seed, nt], dtype=np.int64)
    out["lamsig_params"] = np.array([sig2, lam2])
    out["lamsig_state0"], out["lamsig_goal"] = np.array([0.2, 0.1, -0.4]), goal

    # ---------------- J: the offline harness solve_path (:104-125): ticks until within thresh of the goal
    K, T, seed = 32, 20, 31
    mp = ref.MPPI(horizon=T, samples=K, thresh=0.8)
    np.random.seed(seed)
    mp.solve_path(np.array([0.0, 0.0, 0.0]), np.array([1.0, 0.2, 0.0]))
    out["solve_path_path"] = mp.path.copy()          # [iterations + 1][3]
    out["solve_path_uvec"] = mp.uvec.copy()
    out["solve_path_meta"] = np.array([K, T, seed, mp.path.shape[0] - 1], dtype=np.int64)

    # ---------------- K: sig that is NOT sigma * I.  The reference draws every wheel's noise with sig[0,0] (:143-146)
    # but its stage cost multiplies the FULL matrix, lam * u . sig . eps (:184): a diagonal with two different
    # entries, and a matrix with off-diagonal terms (the reference accepts both).
    K, T, seed, nt = 56, 50, 41, 4
    for tag, sm in (("sigdiag", np.array([[0.9, 0.0], [0.0, 0.5]])), ("sigfull", np.array([[0.8, 0.15], [-0.1, 0.5]]))):
        mp = ref.MPPI(horizon=T, samples=K)
        u0 = nominal_warm(T)
        state, goal = np.array([0.05, -0.1, 0.3]), np.array([0.5, -0.7, -0.2])
        np.random.seed(seed)
        V, eps = mp.get_cost2go(state, u0.copy(), goal, 0.02, sm)
        assert np.array_equal(np.array(eps), np.random.RandomState(seed).normal(0.0, sm[0, 0], (T, 2, K)))
        out[tag + "_c2g_V"] = V.copy()
        out[tag + "_c2g_unew"] = mp.update_action(u0.copy(), eps, V.copy(), sm, 0.02)
        mp.initialize()
        np.random.seed(seed + 1)
        st, states, us = state.copy(), [], []
        for _ in range(nt):
            st = mp.get_path(st, goal, sig=sm, lam=0.02)
            states.append(st.copy())
            us.append(mp.uvec[-1].copy())
        out[tag + "_seq_states"], out[tag + "_seq_u"] = np.array(states), np.array(us)
        out[tag + "_seq_latest_uvec"] = mp.latest_uvec.copy()
        out[tag + "_sig"] = sm
    out["sigmat_meta"] = np.array([K, T, seed, nt], dtype=np.int64)
    out["sigmat_state"], out["sigmat_goal"], out["sigmat_u0"] = state, goal, nominal_warm(T)
    out["sigmat_lam"] = np.array(0.02)

    # ---------------- K (round 3): an ODD horizon.  scipy >= 1.x accepts the even window T - 1 (the scipy of the reference's ROS
    # era raised), so the reference as it runs today takes MPPI(horizon=51): the operator for T = 7, 51, 101 and a closed loop
    for T in (7, 51, 101):
        out["savgol_S_%d" % T] = savgol_filter(np.eye(T), T - 1, 3, axis=1)
    K, T, seed, nt = 24, 51, 8, 6
    mp = ref.MPPI(horizon=T, samples=K)
    np.random.seed(seed)
    st, goal = np.array([0.0, 0.0, 0.0]), np.array([0.3, -0.6, 0.5])
    states, us = [], []
    for _ in range(nt):
        st = mp.get_path(st, goal)
        states.append(st.copy())
        us.append(mp.uvec[-1].copy())
    out["odd_seq_states"], out["odd_seq_u"], out["odd_seq_latest_uvec"] = np.array(states), np.array(us), mp.latest_uvec.copy()
    out["odd_seq_meta"] = np.array([K, T, seed, nt], dtype=np.int64)
    out["odd_seq_goal"] = goal

    # ---------------- L (round 3): uvec_init is an instance attribute too (:65): initialize() loads it into latest_uvec (:81) and
    # every receding-horizon shift appends uvec_init[:, 0] (:101) -- a non-zero one
    K, T, seed, nt = 20, 50, 17, 6
    mp = ref.MPPI(horizon=T, samples=K)
    mp.uvec_init = np.array([np.linspace(0.8, -0.4, T), np.linspace(-0.3, 0.9, T)])
    mp.initialize()
    np.random.seed(seed)
    st, goal = np.array([0.05, 0.0, 0.2]), np.array([0.0, -1.0, 0.0])
    states, us, lat = [], [], []
    for _ in range(nt):
        st = mp.get_path(st, goal)
        states.append(st.copy())
        us.append(mp.uvec[-1].copy())
        lat.append(mp.latest_uvec.copy())
    out["init_seq_states"], out["init_seq_u"], out["init_seq_latest_uvec"] = np.array(states), np.array(us), np.array(lat)
    out["init_seq_meta"] = np.array([K, T, seed, nt], dtype=np.int64)
    out["init_seq_state0"], out["init_seq_goal"], out["init_seq_uvec_init"] = np.array([0.05, 0.0, 0.2]), goal, mp.uvec_init.copy()

    # ---------------- M (round 3): the smallest sizes the reference itself accepts -- one sample (the softmax of one weight),
    # the shortest horizon scipy's filter takes with polyorder 3 (T = 5: an even window of 4; T = 6: the odd window 5), two and
    # three samples on odd / even horizons, and one sample on the benchmark horizon.  Closed loops of four ticks each.
    edge = [(1, 5, 51), (1, 6, 52), (2, 5, 53), (3, 7, 54), (2, 8, 55), (1, 50, 56)]
    for K, T, seed in edge:
        mp = ref.MPPI(horizon=T, samples=K)
        np.random.seed(seed)
        st, goal = np.array([0.02, -0.01, 0.1]), np.array([0.25, -0.4, -0.3])
        states, us, lat = [], [], []
        for _ in range(4):
            st = mp.get_path(st, goal)
            states.append(st.copy())
            us.append(mp.uvec[-1].copy())
            lat.append(mp.latest_uvec.copy())
        tag = "edge_k%d_t%d" % (K, T)
        out[tag + "_states"], out[tag + "_u"], out[tag + "_latest_uvec"] = np.array(states), np.array(us), np.array(lat)
        out[tag + "_S"] = savgol_filter(np.eye(T), T - 1, 3, axis=1)
    out["edge_meta"] = np.array(edge, dtype=np.int64)
    out["edge_state0"], out["edge_goal"] = np.array([0.02, -0.01, 0.1]), np.array([0.25, -0.4, -0.3])

    # ---------------- N (round 5): Q, R, P1 are plain attributes the reference multiplies WHOLE -- (state - desired).T.dot(Q).dot(...),
    # u.T.dot(R).dot(u) (:181-184), .dot(P1) (:168): matrices with off-diagonal terms.  A symmetric set, and a set that is not
    # symmetric (a quadratic form sees the symmetric part only; the reference accepts either).
    K, T, seed, nt = 48, 50, 61, 4
    Qs = np.array([[1000.0, 150.0, 20.0], [150.0, 800.0, -30.0], [20.0, -30.0, 5.0]])
    Rs = np.array([[1.0, 0.3], [0.3, 2.0]])
    Ps = np.array([[900.0, -200.0, 50.0], [-200.0, 1100.0, 80.0], [50.0, 80.0, 600.0]])
    Qa = Qs + np.array([[0.0, 40.0, -10.0], [-40.0, 0.0, 25.0], [10.0, -25.0, 0.0]])
    Ra = Rs + np.array([[0.0, 0.2], [-0.2, 0.0]])
    Pa = Ps + np.array([[0.0, -60.0, 15.0], [60.0, 0.0, -35.0], [-15.0, 35.0, 0.0]])
    for tag, Qm, Rm, Pm in (("wfull_sym", Qs, Rs, Ps), ("wfull_asym", Qa, Ra, Pa)):
        mp = ref.MPPI(horizon=T, samples=K)
        mp.Q, mp.R, mp.P1 = Qm.copy(), Rm.copy(), Pm.copy()
        u0 = nominal_warm(T)
        state, goal = np.array([0.1, -0.05, 2.9]), np.array([0.4, -1.0, -2.8])
        sig = np.array([[SIG, 0.0], [0.0, SIG]])
        np.random.seed(seed)
        V, eps = mp.get_cost2go(state, u0.copy(), goal, LAM, sig)
        assert np.array_equal(np.array(eps), np.random.RandomState(seed).normal(0.0, SIG, (T, 2, K)))
        out[tag + "_c2g_V"] = V.copy()
        out[tag + "_c2g_unew"] = mp.update_action(u0.copy(), eps, V.copy(), sig, LAM)
        mp.initialize()
        np.random.seed(seed + 1)
        st, states, us = state.copy(), [], []
        for _ in range(nt):
            st = mp.get_path(st, goal)
            states.append(st.copy())
            us.append(mp.uvec[-1].copy())
        out[tag + "_seq_states"], out[tag + "_seq_u"] = np.array(states), np.array(us)
        out[tag + "_seq_latest_uvec"] = mp.latest_uvec.copy()
        out[tag + "_Q"], out[tag + "_R"], out[tag + "_P1"] = Qm, Rm, Pm
    out["wfull_meta"] = np.array([K, T, seed, nt], dtype=np.int64)
    out["wfull_state"], out["wfull_goal"], out["wfull_u0"] = state, goal, nominal_warm(T)

    np.savez_compressed(os.path.join(HERE, "mppi_golden.npz"), **out)
    with open(os.path.join(HERE, "mppi_kat.json"), "w") as f:
        json.dump(kat, f, indent=1, sort_keys=True)
    print("wrote", len(out), "arrays;",
          os.path.getsize(os.path.join(HERE, "mppi_golden.npz")), "bytes")
    for k in sorted(kat):
        print(k, kat[k])


if __name__ == "__main__":
    main()
