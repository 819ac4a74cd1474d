"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/mppi_hip.h declares, fails loudly without a GPU, and its host-only pieces
(config defaults, Savitzky-Golay operator) are right.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(headers=("mppi_hip.h", "mppi_hip_diag.h")):
    names = set()
    for h in headers:
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(mppi_[a-z_0-9]+)\s*\(", src))
    return sorted(names)


def test_header_symbols_all_exported_and_bound():
    from motion_planning_amd import _capi
    lib = _capi.load()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libmppi_hip.so does not export %s" % n
    assert sorted(_capi.SIGNATURES) == names  # the ctypes binding covers exactly the header
    assert lib.mppi_abi_version() == _capi.ABI_VERSION == 5


def test_default_config_is_the_reference_node(kat):
    from motion_planning_amd import _capi
    cfg = _capi.default_config()
    assert (cfg.n_agents, cfg.samples, cfg.horizon) == (1, 10, 100)          # control/src/mppi:62
    assert (cfg.sigma, cfg.lambda_, cfg.floor_w) == (0.9, 0.001, 1e-8)      # :88-89, :193
    assert list(cfg.q) == [1e3, 1e3, 0.0] and list(cfg.r) == [1.0, 1.0] and list(cfg.p1) == [1e3] * 3
    assert cfg.u_max == kat["constants"]["WHEEL_VEL_MAX"]
    assert cfg.wheel_radius == kat["constants"]["WHEEL_RADIUS"]
    assert cfg.wheel_base == kat["constants"]["WHEEL_BASE"]


def test_config_struct_layout_matches_header():
    """sizeof(mppi_config) as the C compiler sees it == the ctypes mirror."""
    import subprocess
    import tempfile
    from motion_planning_amd import _capi
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "sz.c")
        fields = [name for name, _ in _capi.MppiConfig._fields_]
        c_names = [("lambda" if f == "lambda_" else f) for f in fields]
        prog = ('#include <stdio.h>\n#include <stddef.h>\n#include "mppi_hip.h"\nint main(){printf("%zu", sizeof(mppi_config));'
                + "".join('printf(" %%zu", offsetof(mppi_config, %s));' % n for n in c_names)
                + 'printf(" %d %d %d", MPPI_TICK_AUTO, MPPI_TICK_LANES, MPPI_TICK_SCAN);return 0;}')
        open(src, "w").write(prog)
        exe = os.path.join(d, "sz")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        nums = [int(x) for x in subprocess.check_output([exe]).split()]
    assert nums[0] == C.sizeof(_capi.MppiConfig)
    for f, off in zip(fields, nums[1:1 + len(fields)]):  # every field, by name, at the offset the C compiler uses
        assert getattr(_capi.MppiConfig, f).offset == off, f
    assert tuple(nums[-3:]) == (_capi.MPPI_TICK_AUTO, _capi.MPPI_TICK_LANES, _capi.MPPI_TICK_SCAN)


@pytest.mark.parametrize("T", [6, 7, 10, 20, 50, 51, 100, 101, 200])
def test_savgol_operator_native(golden, T):
    from motion_planning_amd.mppi import savgol_matrix
    S = savgol_matrix(T)
    assert np.abs(S - golden["savgol_S_%d" % T]).max() < 2e-12


def test_savgol_accepts_odd_horizons_and_rejects_short_windows():
    """The even window of an odd horizon follows scipy >= 1.x (savgol.hpp; fixtures savgol_S_7 / _51 / _101 above)."""
    from motion_planning_amd.mppi import savgol_matrix
    assert savgol_matrix(51).shape == (51, 51)
    with pytest.raises(ValueError):
        savgol_matrix(4)


def test_host_helpers(kat, orc):
    """The arithmetic the host side keeps: wheelsToTwist, and the two kinematics functions the reference exports at
    module level (control/src/mppi:23-36) -- API parity only, no rollout calls them.  The integrators (`rk4`, `euler`)
    are tokens for the `model=` argument whose calls run the engine's plant kernel (GPU test:
    test_model_tokens_run_the_plant_kernel)."""
    import motion_planning_amd as pkg
    assert np.allclose(pkg.wheels_to_twist([1.0, 2.0]), kat["wheelsToTwist_1_2"], rtol=0, atol=1e-17)
    assert pkg.rk4.name == "rk4" and pkg.euler.name == "euler" and callable(pkg.rk4)
    rs = np.random.RandomState(3)
    for _ in range(20):
        x, u = rs.uniform(-3, 3, 3), rs.uniform(-6, 6, 2)
        assert np.abs(pkg.dd_dynamics(x, u) - orc.dd_dynamics(x, u)).max() < 1e-15
        v, w = u
        assert np.allclose(pkg.unicycle_dynamics(x, u), [np.cos(x[2]) * v, np.sin(x[2]) * v, w], rtol=0, atol=1e-15)
    xs, us = rs.uniform(-3, 3, (3, 5)), rs.uniform(-6, 6, (2, 5))          # the reference calls them on [3, N] / [2, N] too
    assert pkg.dd_dynamics(xs, us).shape == (3, 5)
    assert np.abs(pkg.dd_dynamics(xs, us)[:, 2] - pkg.dd_dynamics(xs[:, 2], us[:, 2])).max() == 0.0
    # a zero-length step is the (wrapped) identity and needs no GPU
    assert np.abs(pkg.rk4(np.array([0.1, 0.2, 0.3]), np.array([1.0, 2.0]), 0.0) - [0.1, 0.2, 0.3]).max() == 0.0
    with pytest.raises(ValueError):
        pkg.rk4(np.zeros(3), np.zeros(2), -0.1)


def test_product_reads_no_test_hooks_from_the_environment():
    """Engine construction is configured by arguments and mppi_set_option only: the Python shell reads no environment
    variable, and the library reads exactly one -- MPPI_SYNC_TIMEOUT_MS, the documented deployment knob (every measurement /
    test switch is a per-handle option)."""
    import glob
    for path in glob.glob(os.path.join(ROOT, "motion_planning_amd", "*.py")):
        src = open(path).read()
        assert "MPPI_TICK_PATH" not in src
        if os.path.basename(path) in ("mppi.py", "_capi.py", "controller.py"):
            assert "os.environ" not in src, path
    reads = []
    for path in glob.glob(os.path.join(ROOT, "motion_planning_amd", "csrc", "*")):
        for m in re.finditer(r'getenv\s*\(\s*"([A-Z_0-9]+)"', open(path).read()):
            reads.append(m.group(1))
    assert reads == ["MPPI_SYNC_TIMEOUT_MS"], reads


def test_create_fails_loudly_without_gpu():
    """No silent CPU path: without a device the engine refuses to exist."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from motion_planning_amd.mppi import Engine
    from motion_planning_amd._capi import MppiError
    with pytest.raises(MppiError) as ei:
        Engine(16, 10)
    assert ei.value.code == -2 and "HIP" in str(ei.value) or "device" in str(ei.value)


def test_config_struct_size_is_checked_before_anything_else():
    """mppi_config.struct_size (ABI 5): the caller's sizeof(mppi_config).  mppi_default_config fills it in; mppi_create copies
    exactly that many bytes over its own defaults (a caller built against an older, shorter struct keeps working when fields are
    appended) and refuses sizes it does not know BEFORE touching a device -- so this runs without a GPU.  The measurement surface
    lives in its own header, which compiles as C next to the product header."""
    import ctypes as C
    import subprocess
    import tempfile
    from motion_planning_amd import _capi
    lib = _capi.load()
    cfg = _capi.default_config()
    # 168 bytes as ABI 5 introduced it + the int64 samples_total appended in round 6
    assert cfg.struct_size == C.sizeof(_capi.MppiConfig) == 176 and _capi.MppiConfig.struct_size.offset == 0
    assert _capi.MppiConfig.samples_total.offset == 168 and cfg.samples_total == 0
    h = C.c_void_p()
    for bad in (0, 167, C.sizeof(_capi.MppiConfig) + 8):
        cfg.struct_size = bad
        assert lib.mppi_create(C.byref(cfg), C.byref(h)) == -1 and h.value is None
        assert b"struct_size" in lib.mppi_last_error(None)
    # a caller compiled against the 168-byte struct: accepted (the appended field takes its default) -- without a GPU the call then
    # fails for the device, not for the struct
    cfg.struct_size = 168
    rc = lib.mppi_create(C.byref(cfg), C.byref(h))
    assert rc in (0, -2) and (rc == 0 or b"struct_size" not in lib.mppi_last_error(None))
    if rc == 0:
        lib.mppi_destroy(h)
    with tempfile.TemporaryDirectory() as d:     # both headers are plain C; the product header does not need the diagnostic one
        src = os.path.join(d, "h.c")
        open(src, "w").write('#include "mppi_hip_diag.h"\nint main(void){ return MPPI_CONFIG_SIZE_V5 + 8 == sizeof(mppi_config) && MPPI_PROBE_MARKS == 30 ? 0 : 1; }\n')
        exe = os.path.join(d, "h")
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        assert subprocess.call([exe]) == 0
    prod = open(os.path.join(ROOT, "include", "mppi_hip.h")).read()
    assert "mppi_kernel_timing" not in prod and "mppi_set_option(" not in prod and "mppi_hip_diag.h" in prod


def test_null_handle_is_an_error_not_a_crash():
    from motion_planning_amd import _capi
    lib = _capi.load()
    assert lib.mppi_synchronize(None) == -1
    assert lib.mppi_destroy(None) == -1
    assert lib.mppi_create(None, None) == -1
    assert lib.mppi_last_error(None) is not None


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under motion_planning_amd/ may reference it."""
    pkg = os.path.join(ROOT, "motion_planning_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(base, f)).read()
                assert "libmppi_oracle" not in txt and "import oracle" not in txt and "from oracle" not in txt, f
                assert "/root/reference" not in txt, f


def test_cpp_node_links_against_the_c_abi(tmp_path):
    """examples/mppi_node.cpp (the compiled, ROS-less caller) builds with g++ against include/mppi_hip.h
    and the in-tree library, and -- on a box without a GPU -- fails loudly instead of computing anything."""
    import os
    import subprocess
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "mppi_node")
    subprocess.run(["make", "-B", "-C", os.path.join(root, "examples"), "OUT=" + exe], check=True, capture_output=True)
    if torch.cuda.is_available():
        return
    out = subprocess.run([exe, "--callbacks", "2"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 2 and "mppi_create" in out.stderr and out.stdout == ""


def test_committed_bench_line_has_the_contract_fields():
    """profiles/r2_bench_c4.json is the line bench.py printed on the GPU box: every field the bench contract names is there,
    the roofline arithmetic (the 8(d) HBM accounting and the VALU-issue roof) is self-consistent and the line names the
    workload of BASELINE.json; the line printed with the driver's own flags agrees with it."""
    import json
    line = json.load(open(os.path.join(ROOT, "profiles", "r2_bench_c4.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["higher_is_better"] is True and line["vs_baseline"] is None
    assert line["unit"] == "rollouts/s" and line["data"] == "synthetic" and line["dtype"] == "f64"
    assert "fp32" in line["dtype_detail"] and "softmax" in line["dtype_detail"]      # the label says what is fp32
    assert "K=1000000 T=50" in line["config"]["workload"] and "model" not in line["config"]
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert roof["actual_bound"] == "valu-issue"
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    # achieved = algorithmic bytes per launch / the event-measured average launch duration
    assert abs(roof["achieved"] - roof["algorithmic_bytes_per_launch"] / (roof["avg_launch_us"] * 1e-6) / 1e9) < 1e-6 * roof["achieved"]
    assert roof["algorithmic_bytes_per_launch"] == 12 * line["config"]["state_steps_per_tick"]
    assert roof["traffic"] < roof["algorithmic_bytes_per_launch"]        # no wasted re-reads: eps is never stored
    valu = roof["valu"]    # wave-instructions x issue cost against 1024 SIMDs at 2.4 GHz
    wave_steps = line["config"]["state_steps_per_tick"] / 64.0
    assert abs(valu["insts_per_launch"] - valu["valu_per_step"] * wave_steps) < 1.0
    t_min = valu["issue_cycles_per_step"] * wave_steps / 1024 / 2.4e9
    assert abs(valu["frac"] - t_min / (roof["avg_launch_us"] * 1e-6)) < 1e-9 and 0.3 < valu["frac"] < 1.0
    assert abs(valu["insts_per_launch_pmc"] / valu["insts_per_launch"] - 1.0) < 0.2   # SQ_INSTS_VALU agrees with the assembly count
    # value = whole-job rollouts per tick / tick time; the tick distribution brackets the mean
    assert abs(line["value"] - line["config"]["samples_total"] / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    assert line["tick_us"]["median"] <= line["tick_us"]["p99"] and abs(line["tick_us"]["median"] / (1e3 * line["ms_per_step"]) - 1) < 0.1
    assert line["f64_storage"]["storage"] == "f64" and line["f64_storage"]["ms_per_step"] > line["ms_per_step"]
    co = line["co_scheduled"]     # two engines on the one GPU: reported beside `value`, never as `value`
    assert co["engines_per_gpu"] == 2 and sum(co["samples_per_engine"]) == line["config"]["samples_total"]
    assert abs(co["value"] - line["config"]["samples_total"] / (co["ms_per_step"] * 1e-3)) < 1e-6 * co["value"]
    assert 0.8 < co["ms_per_step"] / line["ms_per_step"] < 1.1
    cpu = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample", "baseline_md_inputs"):
        assert key in cpu, key
    assert cpu["kind"] in ("reference", "port") and cpu["cores"] >= 1
    assert any(k.startswith("c1_K1000") for k in cpu["baseline_md_inputs"]) and any(k.startswith("c2_K10000") for k in cpu["baseline_md_inputs"])
    drv = json.load(open(os.path.join(ROOT, "profiles", "r2_bench_c4_driver_flags.json")))
    assert drv["steps"] == 20 and drv["warmup"] == 5
    assert abs(drv["ms_per_step"] / line["ms_per_step"] - 1.0) < 0.03      # a 20-step run is steady-state too


def test_committed_round4_bench_line_says_what_bounds_the_kernels():
    """profiles/r4_bench_c4.json (the line bench.py printed on the GPU box with the round's final build): the roofline record
    leads with the roof the dominant kernel is ON -- VALU issue, at the measured clock and at the 2.4 GHz peak --, gives the HBM
    fraction of the rollout and of the update from counter bytes over live durations, keeps SURVEY 8(d)'s accounting figure
    under its own name, and states the tick's floor; all of it self-consistent."""
    import json
    path = os.path.join(ROOT, "profiles", "r4_bench_c4.json")
    if not os.path.exists(path):
        pytest.skip("the round's bench line is committed with the final profile refresh")
    line = json.load(open(path))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["vs_baseline"] is None and line["unit"] == "rollouts/s" and line["dtype"] == "f32"
    assert "K=1000000 T=50" in line["config"]["workload"] and "model" not in line["config"]
    assert abs(line["value"] - line["config"]["samples_total"] / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "hbm", "accounting_8d", "tick_floor_us", "tick_frac", "valu"):
        assert key in roof, key
    assert roof["bound"] == "valu-issue" and roof["unit"] == "G wave-inst/s" and roof["kernel"] == "rollout_pk_kernel"
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9 and 0.5 < roof["frac_at_peak_clock"] <= roof["frac"] * 1.1 < 1.1
    wave_steps = line["config"]["state_steps_per_tick"] / 64.0
    t_min = roof["valu"]["issue_cycles_per_step"] * wave_steps / 1024 / (roof["clock_mhz_under_load"] * 1e6)
    assert abs(roof["frac"] - t_min / (roof["avg_launch_us"] * 1e-6)) < 1e-9 and abs(roof["min_launch_us"] - 1e6 * t_min) < 1e-6
    acc = roof["accounting_8d"]      # 12 B / state-step / kernel over the launch duration, against 8 TB/s
    assert acc["bound"] == "hbm" and acc["peak"] == 8000.0 and acc["algorithmic_bytes_per_launch"] == 12 * line["config"]["state_steps_per_tick"]
    assert abs(acc["achieved"] - acc["algorithmic_bytes_per_launch"] / (roof["avg_launch_us"] * 1e-6) / 1e9) < 1e-6 * acc["achieved"]
    hbm = roof["hbm"]                # what the kernels really move: counter bytes over live durations
    for k in ("rollout", "update"):
        h = hbm[k]
        assert abs(h["achieved"] - h["counter_bytes"] / (h["avg_launch_us"] * 1e-6) / 1e9) < 1e-6 * h["achieved"] and abs(h["frac"] - h["achieved"] / 8000.0) < 1e-12
        assert h["vs_algorithmic"] < 0.5                                  # eps is never stored: about a third of the accounting bytes exist
    assert roof["traffic"] == hbm["rollout"]["counter_bytes"] and 0.15 < hbm["rollout"]["frac"] < 0.4 and 0.5 < hbm["update"]["frac"] < 0.8
    assert abs(hbm["update"]["frac_of_achievable"] - hbm["update"]["achieved"] / 6300.0) < 1e-12
    terms = roof["tick_floor_terms"]
    floor = max(terms["rollout_issue_us"], terms["update_bytes_over_achievable_hbm_us"]) + terms["merge_us_measured"] + terms["finalize_us_measured"]
    assert abs(roof["tick_floor_us"] - floor) < 1e-6 and abs(roof["tick_frac"] - floor / terms["tick_us"]) < 1e-9 and 0.4 < roof["tick_frac"] < 1.0
    assert roof["tick_level"]["accounting_8d"]["frac"] > 0.9           # the contract's tick-level figure: most of its bytes never exist
    one = line["one_engine"]
    assert one["self_check"]["max_abs_diff_u"] <= 1e-10 and one["ms_per_step"] >= line["ms_per_step"] * 0.95
    pack = line["noise_packing_1"]    # the optional 16-bit noise packing next to the default stream, the headline's protocol
    assert pack["one_engine"]["steps"] == line["steps"] and 0.8 * one["rollout_us"] < pack["one_engine"]["rollout_us"] < one["rollout_us"]
    assert pack["co_scheduled"]["ms_per_step"] < one["ms_per_step"] and "not the default" in pack["option"]
    cpu = line["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0


def test_committed_round5_bench_line_is_flat_and_reproducible_from_itself():
    """profiles/r5_bench_c4.json = the line bench.py PRINTED on the GPU box with the round's final build: under
    4 KB, `roofline` and `cpu_baseline` hold scalars only, and every fraction can be recomputed from the scalars next to it -- the VALU
    roof the rollout is on, the HBM fractions of rollout and update from counter bytes over live durations, SURVEY 8(d)'s accounting
    figure per kernel and per tick (which exceeds 1: eps is never stored), the tick's floor; the one-engine and all-fp64 legs and the
    node's blocking call are first-class scalars.  The full nested record is the side file next to it."""
    import json
    path = os.path.join(ROOT, "profiles", "r5_bench_c4.json")
    if not os.path.exists(path):
        pytest.skip("the round's bench line is committed with the final profile refresh")
    text = open(path).read().strip()
    line = json.loads(text)
    assert len(text) <= 4096
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["vs_baseline"] is None and line["unit"] == "rollouts/s" and line["dtype"] == "f32"
    assert "K=1000000 T=50" in line["config"]["workload"] and "model" not in line["config"]
    assert abs(line["value"] - line["config"]["samples_total"] / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    roof, cpu = line["roofline"], line["cpu_baseline"]
    assert all(not isinstance(v, (dict, list)) for v in roof.values()) and all(not isinstance(v, (dict, list)) for v in cpu.values())
    steps = line["config"]["state_steps_per_tick"]
    launch_s, tick_s = roof["avg_launch_us"] * 1e-6, line["ms_per_step"] * 1e-3
    rel = lambda a, b: abs(a - b) <= 2e-6 * abs(b)
    assert roof["bound"] == "valu-issue" and roof["kernel"] == "rollout_pk_kernel"
    t_min = roof["issue_cycles_per_step"] * (steps / 64.0) / 1024 / (roof["clock_mhz_under_load"] * 1e6)       # the roof it is ON
    assert rel(roof["frac"], t_min / launch_s) and rel(roof["min_launch_us"], 1e6 * t_min) and rel(roof["frac"], roof["achieved"] / roof["peak"])
    assert rel(roof["frac_at_peak_clock"], roof["issue_cycles_per_step"] * (steps / 64.0) / 1024 / 2.4e9 / launch_s) and 0.5 < roof["frac_at_peak_clock"] < 1.0
    assert roof["traffic"] == roof["hbm_rollout_bytes"] and rel(roof["hbm_rollout_frac"], roof["hbm_rollout_bytes"] / launch_s / 8e12)
    assert rel(roof["hbm_update_frac"], roof["hbm_update_bytes"] / (roof["hbm_update_us"] * 1e-6) / 8e12)
    assert rel(roof["hbm_update_frac_of_achievable"], roof["hbm_update_frac"] * 8000.0 / 6300.0)
    assert 0.15 < roof["hbm_rollout_frac"] < 0.4 and 0.5 < roof["hbm_update_frac"] < 0.8
    assert roof["accounting_8d_bytes"] == 12 * steps and rel(roof["accounting_8d_frac"], 12 * steps / launch_s / 8e12)    # SURVEY 8(d), per kernel
    assert roof["tick_accounting_8d_bytes"] == 24 * steps and rel(roof["tick_accounting_8d_frac"], 24 * steps / tick_s / 8e12) and roof["tick_accounting_8d_frac"] > 0.9
    assert 0.4 < roof["tick_frac"] < 1.0 and roof["tick_floor_us"] > roof["min_launch_us"]
    assert roof["one_engine_ms"] == line["one_engine_ms"] >= line["ms_per_step"] * 0.95 and roof["f64_ms"] == line["f64_ms"] > line["ms_per_step"]
    assert line["sync_tick_us_median"] > 1e3 * line["ms_per_step"] and line["sync_tick_us_p99"] >= line["sync_tick_us_median"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and abs(cpu["value"] - max(v for k, v in cpu.items() if k.startswith("threads_"))) <= 1e-6 * cpu["value"] and "threads_1_value" in cpu
    full = json.load(open(os.path.join(ROOT, "profiles", "r5_bench_c4_full_record.json")))    # the side file of the same run
    assert full["value"] == line["value"] and full["one_engine"]["self_check"]["max_abs_diff_u"] <= 1e-10
    assert "protocol" in full["f64_storage"] and full["f64_storage"]["steps"] == line["steps"]


def test_committed_round6_bench_line():
    """profiles/r6_bench_c4.json = the line bench.py PRINTED on the GPU box with round 6's final build: the round-5 contract (flat,
    every fraction recomputable from the scalars next to it) plus what round 6 added -- the fp64 leg says which kernel it ran
    (`fused`) and carries its parked-at-goal figure, the blocking call is measured with nothing bracketed (within a few per cent of
    the back-to-back tick + the host's enqueue), `cpu_baseline` carries the 1-thread figure and the quoted Python-reference figure,
    the config says that every size rule was pinned to the whole controller's sample count."""
    import json
    path = os.path.join(ROOT, "profiles", "r6_bench_c4.json")
    if not os.path.exists(path):
        pytest.skip("the round's bench line is committed with the final profile refresh")
    text = open(path).read().strip()
    line = json.loads(text)
    assert len(text) <= 4096
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["vs_baseline"] is None and line["unit"] == "rollouts/s" and line["dtype"] == "f32"
    assert "K=1000000 T=50" in line["config"]["workload"] and "model" not in line["config"]
    assert line["config"]["kernels_pinned_by_samples_total"] == 1000000 and line["config"]["co_shards"] == 2
    assert abs(line["value"] - line["config"]["samples_total"] / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    roof, cpu = line["roofline"], line["cpu_baseline"]
    assert all(not isinstance(v, (dict, list)) for v in roof.values()) and all(not isinstance(v, (dict, list)) for v in cpu.values())
    steps = line["config"]["state_steps_per_tick"]
    launch_s, tick_s = roof["avg_launch_us"] * 1e-6, line["ms_per_step"] * 1e-3
    rel = lambda a, b: abs(a - b) <= 2e-6 * abs(b)
    assert roof["bound"] == "valu-issue" and roof["kernel"] == "rollout_pk_kernel"
    t_min = roof["issue_cycles_per_step"] * (steps / 64.0) / 1024 / (roof["clock_mhz_under_load"] * 1e6)
    assert rel(roof["frac"], t_min / launch_s) and rel(roof["frac"], roof["achieved"] / roof["peak"]) and 0.5 < roof["frac_at_peak_clock"] < 1.0
    assert roof["traffic"] == roof["hbm_rollout_bytes"] and rel(roof["hbm_rollout_frac"], roof["hbm_rollout_bytes"] / launch_s / 8e12)
    assert rel(roof["hbm_update_frac"], roof["hbm_update_bytes"] / (roof["hbm_update_us"] * 1e-6) / 8e12) and 0.5 < roof["hbm_update_frac"] < 0.8
    assert roof["accounting_8d_bytes"] == 12 * steps and rel(roof["accounting_8d_frac"], 12 * steps / launch_s / 8e12)
    assert 0.4 < roof["tick_frac"] < 1.0
    # the reference's own precision: one fused kernel under way (faster than round 5's 0.195 ms), rollout + update parked at the goal
    assert roof["f64_kernel"] == "fused" and roof["f64_ms"] == line["f64_ms"] and line["ms_per_step"] < line["f64_ms"] < 0.190
    assert roof["f64_update_us"] is None and roof["f64_parked_ms"] > roof["f64_ms"]
    # the node's blocking call, nothing bracketed: the back-to-back tick + the host's enqueue of the co-scheduled pair, not 30 us of events
    assert 1e3 * line["ms_per_step"] < line["sync_tick_us_median"] < 1e3 * line["ms_per_step"] + 30.0
    assert cpu["kind"] == "port" and "threads_1_value" in cpu and cpu["python_reference_value"] == 3825.0 and "BASELINE.md" in cpu["python_reference_source"]
    assert cpu["threads_1_value"] < cpu["value"] and cpu["value"] / cpu["python_reference_value"] > 100


def test_python_shell_fast_paths_still_see_every_change():
    """The reference reads Q, R, P1 and uvec_init[:, 0] on every get_path (control/src/mppi:69-73, :101) and grows path / uvec by
    np.concatenate (:97-98).  The shell keeps those semantics on fast paths (a byte comparison of the attributes, buffers that
    double): in-place edits, rebinding, dtype changes and invalid matrices must all still be seen; path / uvec must still be
    plain arrays of the right shape that can be assigned to.  (Host logic only: the engine is a recording stand-in.)"""
    from motion_planning_amd import mppi as M

    class Recorder:
        sigma = lam = None
        def __init__(self): self.calls = []
        def set_weights(self, q, r, p1): self.calls.append(("weights", q.copy(), r.copy(), p1.copy()))
        def set_weight_matrices(self, Q, R, P1): self.calls.append(("matrices", Q.copy(), R.copy(), P1.copy()))
        def set_shift_fill(self, f): self.calls.append(("fill", np.array(f)))
        def set_sig(self, sig, lam): pass
        def set_nominal(self, u): self.calls.append(("nominal",))
        def reset(self): self.calls.append(("reset",))
        def tick(self, state, goal, **kw): return np.array([[1.0, 2.0, 3.0]]) * (len(self.calls) + 1), np.array([[0.5, -0.5]])

    m = M.MPPI.__new__(M.MPPI)
    m.horizon, m.samples, m.thresh, m.rng, m.seed, m._tick, m.dt = 8, 4, 0.05, "philox", 0, 0, 0.125
    m.Q, m.R, m.P1 = np.diag([1e3, 1e3, 0.0]), np.diag([1.0, 1.0]), np.diag([1e3, 1e3, 1e3])
    m.uvec_init = np.zeros((2, 8))
    m._weights_sent = m._weights_raw = None
    m._fill_sent = np.zeros(2)
    m._path_buf = m._uvec_buf = None
    m._eng = eng = Recorder()
    m.start, m.goal = np.zeros(3), np.array([1.0, 0.0, 0.0])
    m.initialize()
    n = lambda kind: sum(c[0] == kind for c in eng.calls)
    m.get_path(m.start, m.goal); m.get_path(m.start, m.goal)
    assert n("weights") == 1 and n("fill") == 0                      # sent once, then the fast path
    m.Q[2, 2] = 5.0                                                   # in place
    m.get_path(m.start, m.goal)
    assert n("weights") == 2 and eng.calls[-1][1][2] == 5.0
    m.R = [[2.0, 0.0], [0.0, 3.0]]                                    # rebound to a list
    m.get_path(m.start, m.goal); m.get_path(m.start, m.goal)
    assert n("weights") == 3 and list(eng.calls[-1][2]) == [2.0, 3.0]
    m.P1 = np.diag([1, 2, 3]).astype(np.int64)                       # another dtype
    m.get_path(m.start, m.goal)
    assert n("weights") == 4 and list(eng.calls[-1][3]) == [1.0, 2.0, 3.0]
    m.Q = np.ones((3, 3))                                             # off-diagonal terms: the whole matrices go (control/src/mppi:181-184 multiplies them)
    m.get_path(m.start, m.goal); m.get_path(m.start, m.goal)
    assert n("matrices") == 1 and n("weights") == 4 and np.all(eng.calls[-1][1] == 1.0) and eng.calls[-1][3].shape == (3, 3)
    m.Q = np.ones((2, 3))
    with pytest.raises(ValueError):
        m.get_path(m.start, m.goal)
    m.Q = np.diag([1e3, 1e3, 0.0])                                    # diagonal again: back to the diagonals' call
    m.get_path(m.start, m.goal)
    assert n("weights") == 5 and n("matrices") == 1
    m.uvec_init[:, 0] = [0.3, -0.2]                                   # in place: the shift fill of the next tick
    m.get_path(m.start, m.goal)
    assert n("fill") == 1 and list(eng.calls[-1][1]) == [0.3, -0.2]
    m.get_path(m.start, m.goal)
    assert n("fill") == 1
    # path / uvec: one row per get_path behind the initial one, plain arrays, assignable
    assert m.path.shape == (12, 3) and m.uvec.shape == (12, 2) and len(m.fin_time) == 12
    assert np.all(m.uvec[-1] == [0.5, -0.5]) and np.all(m.path[0] == 0.0)
    old = m.path
    for _ in range(200):                                              # across several doublings of the buffers
        m.get_path(m.start, m.goal)
    assert m.path.shape == (212, 3) and np.array_equal(m.path[:12], old) and old.shape == (12, 3)
    src = np.zeros((1, 3))
    m.path = src; m.uvec = [[1.0, 2.0]]
    m.get_path(m.start, m.goal)
    assert m.path.shape == (2, 3) and m.uvec.shape == (2, 2) and np.all(m.uvec[0] == [1.0, 2.0])
    # the documented view semantics: assignment copies (the caller's array is never appended to); a held view writes
    # through to the log until the buffer is next re-allocated, and keeps its own rows afterwards
    assert src.shape == (1, 3) and not np.shares_memory(m.path, src)
    held = m.path
    held[0, 0] = 7.0
    assert m.path[0, 0] == 7.0
    for _ in range(100):
        m.get_path(m.start, m.goal)
    assert held.shape == (2, 3) and held[0, 0] == 7.0 and m.path[0, 0] == 7.0 and not np.shares_memory(held, m.path)


def test_bench_started_plain_with_several_gpus_launches_its_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it must start two ranks itself (it used to exit with
    "launch N>1 with torch.distributed.run").  Without a GPU the ranks cannot get further than selecting their device -- what is
    checked here is the launcher: two children of the same command line with RANK 0 / 1, WORLD_SIZE 2 and a 127.0.0.1 rendezvous in
    their environment, a failing rank's exit code handed back, nothing left running.  (The GPU suite runs the real thing.)"""
    import subprocess
    import sys
    import textwrap
    # a stand-in for the interpreter's `torch` that records what each rank saw and fails the way a GPU-less box does
    fake = tmp_path / "torch"
    fake.mkdir()
    (fake / "__init__.py").write_text(textwrap.dedent('''
        import os, sys
        open(os.path.join(os.environ["MPPI_TEST_DIR"], "rank%s" % os.environ.get("RANK", "x")), "w").write(
            " ".join(os.environ.get(k, "-") for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")) + " " + " ".join(sys.argv[1:]))
        raise SystemExit(7 if os.environ.get("RANK") == "1" else 0)
    '''))
    env = dict(os.environ, PYTHONPATH=str(tmp_path), MPPI_TEST_DIR=str(tmp_path))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3"], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 7, (out.returncode, out.stderr[-500:])
    seen = [open(tmp_path / ("rank%d" % r)).read().split() for r in (0, 1)]
    assert [s[0] for s in seen] == ["0", "1"] and [s[1] for s in seen] == ["0", "1"] and all(s[2] == "2" and s[3] == "127.0.0.1" for s in seen)
    assert seen[0][4] == seen[1][4] and int(seen[0][4]) > 0 and all(s[5:] == ["--gpus", "2", "--steps", "3"] for s in seen)
    # --gpus 1 never launches anything: the process itself is the one rank
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], env=dict(env, RANK="x"), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and open(tmp_path / "rankx").read().split()[2] == "-"


def test_samples_total_is_checked_before_any_device_is_touched():
    """mppi_config.samples_total (round 6): 0 = this handle is the whole controller; otherwise the whole controller's samples per
    agent, which must cover this handle's own range [sample_offset, sample_offset + samples) and fit the 32-bit global sample ids.
    Refused by mppi_create before it looks for a device -- so this runs without a GPU."""
    import ctypes as C
    from motion_planning_amd import _capi
    lib = _capi.load()
    h = C.c_void_p()
    for total, offset, samples in ((5, 0, 10), (100, 95, 10), (1 << 33, 0, 10), (-1, 0, 10)):
        cfg = _capi.default_config()
        cfg.samples, cfg.horizon, cfg.sample_offset, cfg.samples_total = samples, 50, offset, total
        assert lib.mppi_create(C.byref(cfg), C.byref(h)) == -1 and h.value is None, (total, offset, samples)
        assert b"samples_total" in lib.mppi_last_error(None)
    cfg = _capi.default_config()
    cfg.samples, cfg.horizon, cfg.sample_offset, cfg.samples_total = 10, 50, 90, 100      # a valid share: past this check (fails for the device here)
    rc = lib.mppi_create(C.byref(cfg), C.byref(h))
    assert rc in (0, -2) and (rc == 0 or b"samples_total" not in lib.mppi_last_error(None))
    if rc == 0:
        lib.mppi_destroy(h)
