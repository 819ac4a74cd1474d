"""Pins the CPU oracle (oracle/mppi_oracle.c) to golden vectors produced by the REAL
reference (tests/golden/make_golden.py, run in the build container).  CPU only."""
import numpy as np
import pytest

SIG, LAM = 0.9, 0.001
C2G = ["c2g_zero", "c2g_warm", "c2g_wrap", "c2g_clip", "c2g_tiny"]


def test_kat_kinematics(orc, kat):
    assert np.allclose(orc.dd_dynamics([0.3, -0.2, 0.7], [1.25, -0.5]), kat["dd_dynamics"], rtol=0, atol=1e-17)
    assert np.allclose(orc.rk4([0.3, -0.2, 0.7], [1.25, -0.5], 0.01), kat["rk4_plain"], rtol=0, atol=1e-16)
    # theta wrap branch of rk4 (control/src/mppi:52-53)
    assert np.allclose(orc.rk4([0, 0, 3.1], [-6.35492, 6.35492], 0.02), kat["rk4_wrap"], rtol=0, atol=1e-15)
    assert np.allclose(orc.wheels_to_twist([1.0, 2.0]), kat["wheelsToTwist_1_2"], rtol=0, atol=1e-17)
    p = orc.default_params()
    assert p.u_max == kat["constants"]["WHEEL_VEL_MAX"]
    assert p.wheel_radius == kat["constants"]["WHEEL_RADIUS"]
    assert p.wheel_base == kat["constants"]["WHEEL_BASE"]


@pytest.mark.parametrize("T", [6, 7, 10, 20, 50, 51, 100, 101, 200])
def test_savgol_operator_matches_scipy(orc, golden, T):
    S = orc.savgol_matrix(T)
    ref = golden["savgol_S_%d" % T]
    assert np.abs(S - ref).max() < 2e-12
    # rank-6 operator, rows of an interpolating filter sum to 1 column-wise
    assert np.linalg.matrix_rank(S, tol=1e-9) <= 8
    assert np.allclose(S.sum(axis=0), 1.0, atol=1e-12)


def test_savgol_apply(orc, golden):
    S = orc.savgol_matrix(50)
    assert np.abs(golden["savgol_in_50"] @ S - golden["savgol_out_50"]).max() < 1e-12


def test_savgol_window_too_short_rejected(orc):
    """Odd horizons (even windows) are accepted the way scipy >= 1.x accepts them (fixtures savgol_S_7 / _51 / _101);
    a window that cannot hold a cubic is still refused."""
    with pytest.raises(ValueError):
        orc.savgol_matrix(4)
    assert orc.savgol_matrix(51).shape == (51, 51)


def test_odd_horizon_closed_loop(orc, golden):
    """MPPI(horizon=51) of the reference as it runs on today's scipy: six closed-loop ticks."""
    K, T, seed, nt = [int(x) for x in golden["odd_seq_meta"]]
    noise = orc.reference_noise(seed, SIG, T, K, n_ticks=nt)
    st, lat = np.zeros(3), np.zeros((2, T))
    for i in range(nt):
        st, ua, lat = orc.get_path(st, golden["odd_seq_goal"], lat, noise[i], LAM, SIG)
        assert np.abs(st - golden["odd_seq_states"][i]).max() < 1e-10, i
        assert np.abs(ua - golden["odd_seq_u"][i]).max() < 1e-9, i
    assert np.abs(lat - golden["odd_seq_latest_uvec"]).max() < 1e-9


def test_smallest_sizes_closed_loop(orc, golden):
    """Golden section M: one, two, three samples; horizons 5 (even filter window 4), 6, 7, 8 and 50 -- four closed-loop ticks of
    the reference each, and the filter operator scipy builds for those horizons."""
    for K, T, seed in [[int(x) for x in row] for row in golden["edge_meta"]]:
        tag = "edge_k%d_t%d" % (K, T)
        assert np.abs(orc.savgol_matrix(T) - golden[tag + "_S"]).max() < 2e-12, tag
        noise = orc.reference_noise(seed, SIG, T, K, n_ticks=4)
        st, lat = golden["edge_state0"].copy(), np.zeros((2, T))
        for i in range(4):
            st, ua, lat = orc.get_path(st, golden["edge_goal"], lat, noise[i], LAM, SIG)
            assert np.abs(st - golden[tag + "_states"][i]).max() < 1e-10, (tag, i)
            assert np.abs(ua - golden[tag + "_u"][i]).max() < 1e-9, (tag, i)
            assert np.abs(lat - golden[tag + "_latest_uvec"][i]).max() < 1e-9, (tag, i)


def test_nonzero_uvec_init_closed_loop(orc, golden):
    """uvec_init != 0 (control/src/mppi:65): initialize() loads it (:81), every shift appends uvec_init[:, 0] (:101)."""
    K, T, seed, nt = [int(x) for x in golden["init_seq_meta"]]
    noise = orc.reference_noise(seed, SIG, T, K, n_ticks=nt)
    init = golden["init_seq_uvec_init"]
    p = orc.default_params()
    p.shift_fill[0], p.shift_fill[1] = init[0, 0], init[1, 0]
    st, lat = golden["init_seq_state0"].copy(), init.copy()
    for i in range(nt):
        st, ua, lat = orc.get_path(st, golden["init_seq_goal"], lat, noise[i], LAM, SIG, params=p)
        assert np.abs(st - golden["init_seq_states"][i]).max() < 1e-10, i
        assert np.abs(ua - golden["init_seq_u"][i]).max() < 1e-9, i
        assert np.abs(lat - golden["init_seq_latest_uvec"][i]).max() < 1e-9, i
        assert lat[0, -1] == init[0, 0] and lat[1, -1] == init[1, 0]


@pytest.mark.parametrize("name", C2G)
def test_get_cost2go(orc, golden, name):
    K, T, seed = golden[name + "_meta"]
    eps = orc.reference_noise(int(seed), SIG, int(T), int(K))
    V = orc.get_cost2go(golden[name + "_state"], golden[name + "_u0"], golden[name + "_goal"], LAM, SIG, eps)
    ref = golden[name + "_V"]
    assert V.shape == ref.shape
    assert np.abs(V - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())
    assert np.abs(V - ref).max() < 1e-9


@pytest.mark.parametrize("name", C2G)
def test_update_action(orc, golden, name):
    K, T, seed = golden[name + "_meta"]
    eps = orc.reference_noise(int(seed), SIG, int(T), int(K))
    # from the reference's own V (isolates update_action) ...
    u = orc.update_action(golden[name + "_u0"], eps, golden[name + "_V"], LAM, S=golden["savgol_S_%d" % T]
                          if "savgol_S_%d" % T in golden.files else None)
    assert np.abs(u - golden[name + "_unew"]).max() < 1e-10
    # ... and end to end through the oracle's V and its own savgol operator
    V = orc.get_cost2go(golden[name + "_state"], golden[name + "_u0"], golden[name + "_goal"], LAM, SIG, eps)
    u2 = orc.update_action(golden[name + "_u0"], eps, V, LAM)
    assert np.abs(u2 - golden[name + "_unew"]).max() < 1e-9


@pytest.mark.parametrize("name", ["seq_park", "seq_wp"])
def test_closed_loop_ticks(orc, golden, name):
    K, T, seed, nt = [int(x) for x in golden[name + "_meta"]]
    noise = orc.reference_noise(seed, SIG, T, K, n_ticks=nt)
    st = golden[name + "_state0"].copy()
    lat = np.zeros((2, T))
    for i in range(nt):
        st, ua, lat = orc.get_path(st, golden[name + "_goal"], lat, noise[i], LAM, SIG)
        assert np.abs(st - golden[name + "_states"][i]).max() < 1e-10, i
        assert np.abs(ua - golden[name + "_u"][i]).max() < 1e-9, i
        assert np.abs(lat - golden[name + "_latest_uvec"][i]).max() < 1e-9, i


def test_default_node_ticks(orc, kat):
    """MPPI() as shipped (K=10, T=100; control/src/mppi:62) two ticks, SURVEY 8c known answers."""
    noise = orc.reference_noise(0, SIG, 100, 10, n_ticks=2)
    lat = np.zeros((2, 100))
    st, ua, lat = orc.get_path([0, 0, 0], [0, -1, 0], lat, noise[0], LAM, SIG)
    assert np.allclose(st, kat["default_tick1_state"], rtol=0, atol=1e-13)
    assert np.allclose(ua, kat["default_tick1_u"], rtol=0, atol=1e-10)
    assert abs(lat.sum() - kat["default_tick1_sum_latest_uvec"]) < 1e-8
    st, ua, lat = orc.get_path(st, [0, -1, 0], lat, noise[1], LAM, SIG)
    assert np.allclose(st, kat["default_tick2_state"], rtol=0, atol=1e-13)
    assert np.allclose(ua, kat["default_tick2_u"], rtol=0, atol=1e-10)


def test_h50k64_tick(orc, kat):
    noise = orc.reference_noise(0, SIG, 50, 64)
    st, ua, _ = orc.get_path([0, 0, 0], [1, 0, 0], np.zeros((2, 50)), noise, LAM, SIG)
    assert np.allclose(st, kat["h50k64_tick1_state"], rtol=0, atol=1e-13)
    assert np.allclose(ua, kat["h50k64_tick1_u"], rtol=0, atol=1e-10)


@pytest.mark.parametrize("nom", ["zero", "warm"])
def test_config1_k1000(orc, golden, nom):
    """BASELINE config 1: K=1000, T=50 parallel park, seed 0."""
    K, T = 1000, 50
    eps = orc.reference_noise(0, SIG, T, K)
    u0 = np.zeros((2, T)) if nom == "zero" else np.array([np.linspace(-2, 1, T), np.linspace(1.5, -1, T)])
    V = orc.get_cost2go([0, 0, 0], u0, [0, -1, 0], LAM, SIG, eps)
    assert np.abs(V[:, ::100] - golden["c1_%s_Vcols" % nom]).max() < 1e-9
    assert np.abs(V.min(axis=1) - golden["c1_%s_Vmin_t" % nom]).max() < 1e-9
    assert np.abs(V.sum(axis=1) - golden["c1_%s_Vsum_t" % nom]).max() < 1e-6
    assert np.abs(orc.update_action(u0, eps, V, LAM) - golden["c1_%s_unew" % nom]).max() < 1e-9
    st, ua, _ = orc.get_path([0, 0, 0], [0, -1, 0], u0, eps, LAM, SIG)
    assert np.abs(st - golden["c1_%s_next_state" % nom]).max() < 1e-12
    assert np.abs(ua - golden["c1_%s_u_applied" % nom]).max() < 1e-9


def test_shard_merge_equals_update(orc, golden):
    """splitting K over G shards + merge == update_action's increment (incl. the 1e-8 floor)."""
    name = "c2g_clip"
    K, T, seed = [int(x) for x in golden[name + "_meta"]]
    eps = orc.reference_noise(seed, SIG, T, K)
    V = golden[name + "_V"]
    full = orc.merge_partials(orc.shard_partials(eps, V, 0, K, LAM)[None], [K], LAM)
    for G in (2, 4, 8):
        b = np.linspace(0, K, G + 1).astype(int)
        parts = np.stack([orc.shard_partials(eps, V, b[g], b[g + 1], LAM) for g in range(G)])
        du = orc.merge_partials(parts, np.diff(b), LAM)
        assert np.abs(du - full).max() < 1e-13
    # and the increment is what update_action adds before clip/filter
    Vc = V - V.min(axis=1, keepdims=True)
    w = np.exp(-Vc / LAM) + 1e-8
    w /= w.sum(axis=1, keepdims=True)
    du_ref = np.einsum("tck,tk->ct", eps, w)
    assert np.abs(full - du_ref).max() < 1e-13


def test_philox_known_answers(orc):
    """Random123 known-answer vectors for Philox4x32-10 (Salmon et al., SC'11 distribution kat_vectors)."""
    assert orc.philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert orc.philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert orc.philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
                             [0xa4093822, 0x299f31d0]) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_philox_noise_statistics(orc):
    e = orc.philox_noise(seed=0, agent=0, tick=0, k_off=0, K_local=20000, T=6, sigma=0.9)
    assert abs(e.mean()) < 0.01 and abs(e.std() - 0.9) < 0.01
    # shard-count invariance: sample ids are global
    a = orc.philox_noise(3, 1, 2, 0, 64, 7, 0.9)
    b = orc.philox_noise(3, 1, 2, 32, 32, 7, 0.9)
    assert np.array_equal(a[:, :, 32:], b)


def test_euler_unicycle_model(orc, golden, kat):
    """The `model=euler` constructor path (control/src/mppi:33-36, :57-58): unicycle kinematics, no wrap."""
    p = orc.default_params()
    p.model = 1
    assert np.allclose(orc.euler([0.3, -0.2, 3.1], [1.25, -0.5], 0.25), kat["euler_step"], rtol=0, atol=1e-16)
    K, T, seed = [int(x) for x in golden["euler_c2g_meta"]]
    eps = orc.reference_noise(seed, SIG, T, K)
    V = orc.get_cost2go(golden["euler_c2g_state"], golden["euler_c2g_u0"], golden["euler_c2g_goal"], LAM, SIG, eps, params=p)
    assert np.abs(V - golden["euler_c2g_V"]).max() < 1e-9
    assert np.abs(orc.update_action(golden["euler_c2g_u0"], eps, V, LAM, params=p) - golden["euler_c2g_unew"]).max() < 1e-9
    K, T, seed, nt = [int(x) for x in golden["euler_seq_meta"]]
    noise = orc.reference_noise(seed, SIG, T, K, n_ticks=nt)
    st, lat = np.array([0.0, 0.0, 2.5]), np.zeros((2, T))
    for i in range(nt):
        st, ua, lat = orc.get_path(st, [-0.5, 0.4, 3.0], lat, noise[i], LAM, SIG, params=p)
        assert np.abs(st - golden["euler_seq_states"][i]).max() < 1e-10
        assert np.abs(ua - golden["euler_seq_u"][i]).max() < 1e-9


def test_general_weights_golden(orc, golden):
    """Q, R, P1 other than the node's constants, pinned by the reference itself (its MPPI instance with the
    attributes of control/src/mppi:69-73 overwritten; tests/golden/make_golden.py section H)."""
    p = orc.default_params()
    p.q[:], p.r[:], p.p1[:] = golden["wts_q"], golden["wts_r"], golden["wts_p1"]
    K, T, seed = [int(x) for x in golden["wts_c2g_meta"]]
    eps = orc.reference_noise(seed, SIG, T, K)
    state, goal, u0 = golden["wts_c2g_state"], golden["wts_c2g_goal"], golden["wts_c2g_u0"]
    V = orc.get_cost2go(state, u0, goal, LAM, SIG, eps, params=p)
    assert np.abs(V - golden["wts_c2g_V"]).max() < 1e-9 * np.abs(golden["wts_c2g_V"]).max()
    assert np.abs(orc.update_action(u0, eps, V, LAM, params=p) - golden["wts_c2g_unew"]).max() < 1e-9
    K, T, seed, nt = [int(x) for x in golden["wts_seq_meta"]]
    noise = orc.reference_noise(seed, SIG, T, K, n_ticks=nt)
    st, lat = state.copy(), np.zeros((2, T))
    for i in range(nt):
        st, ua, lat = orc.get_path(st, goal, lat, noise[i], LAM, SIG, params=p)
        assert np.abs(st - golden["wts_seq_states"][i]).max() < 1e-10
        assert np.abs(ua - golden["wts_seq_u"][i]).max() < 1e-9


@pytest.mark.parametrize("tag", ["wfull_sym", "wfull_asym"])
def test_full_weight_matrices_golden(orc, golden, tag):
    """Q, R, P1 with OFF-DIAGONAL terms -- the reference multiplies the whole matrices (control/src/mppi:168, :181-184) --,
    a symmetric set and one that is not (golden section N: the reference's MPPI instance with its attributes overwritten)."""
    p = orc.set_weight_matrices(orc.default_params(), golden[tag + "_Q"], golden[tag + "_R"], golden[tag + "_P1"])
    K, T, seed, nt = [int(x) for x in golden["wfull_meta"]]
    eps = orc.reference_noise(seed, SIG, T, K)
    state, goal, u0 = golden["wfull_state"], golden["wfull_goal"], golden["wfull_u0"]
    V = orc.get_cost2go(state, u0, goal, LAM, SIG, eps, params=p)
    Vg = golden[tag + "_c2g_V"]
    assert np.abs(V - Vg).max() < 1e-12 * np.abs(Vg).max()
    assert np.abs(orc.update_action(u0, eps, V, LAM, params=p) - golden[tag + "_c2g_unew"]).max() < 1e-9
    noise = orc.reference_noise(seed + 1, SIG, T, K, n_ticks=nt)
    st, lat = state.copy(), np.zeros((2, T))
    for i in range(nt):
        st, ua, lat = orc.get_path(st, goal, lat, noise[i], LAM, SIG, params=p)
        assert np.abs(st - golden[tag + "_seq_states"][i]).max() < 1e-10
        assert np.abs(ua - golden[tag + "_seq_u"][i]).max() < 1e-9
    assert np.abs(lat - golden[tag + "_seq_latest_uvec"]).max() < 1e-9
    # the off-diagonal terms matter: the diagonals alone give another V
    pd = orc.default_params()
    pd.q[:], pd.r[:], pd.p1[:] = np.diag(golden[tag + "_Q"]), np.diag(golden[tag + "_R"]), np.diag(golden[tag + "_P1"])
    assert np.abs(orc.get_cost2go(state, u0, goal, LAM, SIG, eps, params=pd) - Vg).max() > 1.0


def test_other_sigma_and_lambda_golden(orc, golden):
    """get_path with the sig / lam arguments the node never changes (control/src/mppi:88-89), golden section I."""
    K, T, seed, nt = [int(x) for x in golden["lamsig_seq_meta"]]
    sig2, lam2 = [float(x) for x in golden["lamsig_params"]]
    noise = orc.reference_noise(seed, sig2, T, K, n_ticks=nt)
    st, lat = golden["lamsig_state0"].copy(), np.zeros((2, T))
    for i in range(nt):
        st, ua, lat = orc.get_path(st, golden["lamsig_goal"], lat, noise[i], lam2, sig2)
        assert np.abs(st - golden["lamsig_seq_states"][i]).max() < 1e-10
        assert np.abs(ua - golden["lamsig_seq_u"][i]).max() < 1e-9
    assert np.abs(lat - golden["lamsig_seq_latest_uvec"]).max() < 1e-9


def test_solve_path_golden(orc, golden):
    """The offline harness (control/src/mppi:104-125, sig = I, lam = 0.01): tick until within thresh of the goal."""
    K, T, seed, n_it = [int(x) for x in golden["solve_path_meta"]]
    path = golden["solve_path_path"]
    noise = orc.reference_noise(seed, 1.0, T, K, n_ticks=n_it)
    st, goal, lat = np.zeros(3), np.array([1.0, 0.2, 0.0]), np.zeros((2, T))
    i = 0
    while np.linalg.norm(st[:2] - goal[:2]) > 0.8:
        st, ua, lat = orc.get_path(st, goal, lat, noise[i], 0.01, 1.0)
        i += 1
        assert np.abs(st - path[i]).max() < 1e-10 and np.abs(ua - golden["solve_path_uvec"][i]).max() < 1e-9
    assert i == n_it


def test_obstacle_grid_extension_is_off_by_default(orc, golden):
    """The obstacle-grid stage cost is NOT in the reference: weight 0 / no grid must
    leave the golden results untouched, a weighted grid must add exactly weight*value/100 per step."""
    name = "c2g_warm"
    K, T, seed = [int(x) for x in golden[name + "_meta"]]
    eps = orc.reference_noise(seed, SIG, T, K)
    p = orc.default_params()
    cells = (50 * ((np.arange(120)[:, None] + 2 * np.arange(160)[None, :]) % 3)).astype(np.int8)  # 0/50/100 stripes
    orc.set_obstacle_grid(p, cells, 0.0125, (-0.5, -0.9), 0.0)
    args = (golden[name + "_state"], golden[name + "_u0"], golden[name + "_goal"], LAM, SIG, eps)
    assert np.array_equal(orc.get_cost2go(*args, params=p), orc.get_cost2go(*args))
    orc.set_obstacle_grid(p, cells, 0.0125, (-0.5, -0.9), 300.0)
    V, c, xT = orc.get_cost2go(*args, params=p, want_costs=True)
    V0, c0, _ = orc.get_cost2go(*args, want_costs=True)
    extra = c - c0
    assert extra.min() >= 0 and set(np.unique(np.round(extra, 9))) <= {0.0, 150.0, 300.0} and extra.max() > 0


@pytest.mark.parametrize("tag", ["sigdiag", "sigfull"])
def test_sig_matrix_not_isotropic(orc, golden, tag):
    """sig != sigma * I (golden section K): the noise is drawn with sig[0,0] for both wheels
    (control/src/mppi:143-146) while the stage cost uses the full matrix, lam * u . sig . eps (:184)."""
    K, T, seed, nt = [int(x) for x in golden["sigmat_meta"]]
    sm, lam = golden[tag + "_sig"], float(golden["sigmat_lam"])
    p = orc.set_sig_matrix(orc.default_params(), sm)
    state, goal, u0 = golden["sigmat_state"], golden["sigmat_goal"], golden["sigmat_u0"]
    eps = orc.reference_noise(seed, sm[0, 0], T, K)
    V = orc.get_cost2go(state, u0, goal, lam, sm[0, 0], eps, params=p)
    assert np.abs(V - golden[tag + "_c2g_V"]).max() < 1e-9
    u = orc.update_action(u0, eps, V, lam, params=p)
    assert np.abs(u - golden[tag + "_c2g_unew"]).max() < 1e-9
    # the isotropic shortcut would be wrong here: the goldens really exercise the matrix
    Viso = orc.get_cost2go(state, u0, goal, lam, sm[0, 0], eps)
    assert np.abs(Viso - golden[tag + "_c2g_V"]).max() > 1e-4
    noise = orc.reference_noise(seed + 1, sm[0, 0], T, K, n_ticks=nt)
    st, lat = state.copy(), np.zeros((2, T))
    for i in range(nt):
        st, ua, lat = orc.get_path(st, goal, lat, noise[i], lam, sm[0, 0], params=p)
        assert np.abs(st - golden[tag + "_seq_states"][i]).max() < 1e-10, i
        assert np.abs(ua - golden[tag + "_seq_u"][i]).max() < 1e-9, i
    assert np.abs(lat - golden[tag + "_seq_latest_uvec"]).max() < 1e-9


@pytest.mark.parametrize("packing", [0, 1, 2])
def test_device_noise_twins_use_the_philox_words_as_documented(orc, packing):
    """The CPU twins of the engine's three noise packings (include/mppi_hip.h, option "noise_packing") restated once more in numpy
    from the raw Philox4x32-10 words: which bits of which call serve which step, the Box-Muller radius each packing can reach, and
    -- on 2 x 10^5 draws -- zero mean, variance sigma^2, uncorrelated wheels.  (The GPU suite ties the device noise to these twins,
    and packing 2 to hiprand_normal4() bit for bit.)"""
    sigma, seed, agent, tick, k_off, K, T = 0.9, (7 << 32) | 12345, 3, 41, 1000, 96, 23
    eps = orc.philox_noise(seed, agent, tick, k_off, K, T, sigma, packing=packing)
    assert eps.shape == (T, 2, K) and np.isfinite(eps).all()
    spd = {0: 3, 1: 4, 2: 2}[packing]
    key = [seed & 0xFFFFFFFF, seed >> 32]
    for k in (0, 1, K - 1):
        for t in (0, 1, 2, 3, 4, 5, 7, T - 1):
            o = orc.philox4x32_10([k_off + k, t // spd, tick, agent], key)
            j = t % spd
            if packing == 0:
                a = [o[0] >> 11, o[2] >> 11, ((o[0] & 0x7FF) << 10) | ((o[1] & 0x7FF) >> 1)][j]
                b = [o[1] >> 11, o[3] >> 11, ((o[2] & 0x7FF) << 10) | ((o[3] & 0x7FF) >> 1)][j]
                u1, u2 = (a + 0.5) / 2**21, b / 2**21
            elif packing == 1:
                u1, u2 = ((o[j] & 0xFFFF) + 0.5) / 2**16, (o[j] >> 16) / 2**16
            else:   # rocRAND's box_muller: u = 2^-32 + x 2^-32, v = 2 pi 2^-32 (1 + y), (sin v, cos v) sqrt(-2 ln u), in fp32
                x, y = np.float32(o[2 * j]), np.float32(o[2 * j + 1])
                u = np.float32(np.float64(x) * np.float64(np.float32(2.3283064e-10)) + np.float64(np.float32(2.3283064e-10)))
                v = np.float32(np.float64(y) * np.float64(np.float32(1.46291807e-09)) + np.float64(np.float32(1.46291807e-09)))
                s = np.sqrt(-2.0 * np.log(np.float64(u)))
                want = np.array([np.sin(np.float64(v)) * s, np.cos(np.float64(v)) * s]) * sigma
                assert np.abs(eps[t, :, k] - want).max() < 2e-6 * max(1.0, s)
                continue
            r = sigma * np.sqrt(-2.0 * np.log(u1))
            want = np.array([r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)])
            assert np.abs(eps[t, :, k] - want).max() < 1e-6
    big = orc.philox_noise(seed, 0, 0, 0, 4096, 25, sigma, packing=packing)          # 204 800 normals
    radius = {0: 5.53, 1: 4.86, 2: 6.67}[packing]
    assert np.abs(big).max() <= radius * sigma * 1.000001
    n = big.size
    assert abs(big.mean()) < 4 * sigma / np.sqrt(n)
    assert abs(big.var() / sigma**2 - 1.0) < 0.02
    assert abs(np.corrcoef(big[:, 0].ravel(), big[:, 1].ravel())[0, 1]) < 0.02


@pytest.mark.parametrize("name", ["c2g_zero", "c2g_warm", "c2g_wrap", "c2g_clip", "c2g_tiny"])
def test_update_action_side_effects_golden(orc, golden, name):
    """update_action mutates its ARGUMENTS (control/src/mppi:189, :196-199): every row of value_fcn loses its minimum, uvec receives
    the weighted noise and the first clip, both in place; what it returns (the filtered sequence) is a new array.  The oracle's C
    restatement does the same to its buffers: pinned to what the imported reference left in the arrays it was given."""
    K, T, seed = [int(x) for x in golden[name + "_meta"]]
    eps = orc.reference_noise(seed, 0.9, T, K)
    out, u_after, V_after = orc.update_action(golden[name + "_u0"], eps, golden[name + "_V"], 0.001, want_inplace=True)
    assert np.abs(out - golden[name + "_unew"]).max() < 1e-12
    assert np.abs(u_after - golden[name + "_u_inplace"]).max() < 1e-12
    assert np.array_equal(V_after, golden[name + "_V_inplace"])
