#!/usr/bin/env python3
"""bench.py -- MPPI rollouts/s per control tick on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c4|c2|c3|c5] [--storage f32|f64]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full control tick (MPPI.get_path, control/src/mppi:85-102) on synthetic inputs, device
Philox noise, closed loop on the model (the predicted state feeds the next tick), everything resident in
HBM: nominal baseline -> rollout+cost -> per-timestep softmax partials -> [exchange of the [A][T][8]
partials when K is sharded] -> control update, clip, Savitzky-Golay, clip, plant step, shift.

Default workload = BASELINE config 4, the one the north-star target is quoted on: parallel park,
K = 1 000 000 rollouts, T = 50, K split over the N GPUs (strong scaling: the north star "2/4/8-GPU runs
split K").  It fits one GPU, so N = 1 runs all of it.  --workload c2 / c5 select other BASELINE configs;
--workload c3 is config 3 as SURVEY 8d-3 specifies it: the node shell driven through the five waypoints of
control/config/waypoints.yaml with blocking ticks (goal switches need the pose on the host every tick).
Prints ONE JSON line on rank 0.
"""
import argparse
import gc
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
CLOCK_PEAK_HZ = 2.4e9  # max shader clock, same guide
N_SIMD = 1024          # 256 CUs x 4 SIMDs
BYTES_PER_STEP_PER_KERNEL = 12  # eps 2 x fp32 + V 1 x fp32, written once (rollout) / read once (update): SURVEY 8(d)
PROFILE_ROUND = "r6"
PYTHON_REFERENCE_ROLLOUTS_PER_S = 3825.0   # MPPI.get_path, K = 1000, T = 50, one core of the survey container (quoted)
HBM_ACHIEVABLE_GBS = 6300.0  # what a streaming read reaches on this part (MI355X_MICROARCH.md, HBM section)

WORKLOADS = {
    # name: (description, agents, K_total, T, goal)
    "c4": ("parallel-park K=1000000 T=50 (BASELINE config 4; K split over the GPUs)", 1, 1000000, 50, [0.0, -1.0, 0.0]),
    "c2": ("parallel-park K=10000 T=50 (BASELINE config 2)", 1, 10000, 50, [0.0, -1.0, 0.0]),
    "c3": ("pentagon waypoint-follow K=100000 T=100 (BASELINE config 3: the node shell through the five waypoints "
           "of control/config/waypoints.yaml, blocking ticks)", 1, 100000, 100, [1.0, 0.0, 0.0]),
    "c5": ("64 agents x K=16384 T=50 (BASELINE config 5; agents split over the GPUs)", 64, 16384, 50, None),
}
PENTAGON = [[1, 0], [2, 1], [1, 2], [0, 2], [0, 0]]   # control/config/waypoints.yaml:1


def nominal_warm(T):
    return np.array([np.linspace(-2.0, 1.0, T), np.linspace(1.5, -1.0, T)])


def host_cores():
    """Cores this process may really use: scheduler affinity capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.999)))
    except (OSError, ValueError):
        try:  # cgroup v1
            quota = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = max(1, min(n, int(quota / period + 0.999)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(T, goal, budget_s=14.0):
    """The oracle (C restatement of the reference loop, oracle/mppi_oracle.c) timed on this box's host cores.
    `value`: a bounded sample of the SAME workload (K_cpu rollouts of the same T, goal and nominal), on 1 thread
    (the scalar port) and with OpenMP over K on every usable core; the better one is `value`.
    `baseline_md_inputs`: BASELINE.md section 3's own inputs -- config 1 (K = 1000) and config 2 (K = 10 000), T = 50,
    noise RandomState(0), zero and warm nominal controls -- median and p99 tick time at 1 thread and at all cores."""
    from oracle import oracle as orc
    orc.build()
    cores = host_cores()
    K_cpu = 100000
    eps = np.random.RandomState(0).normal(0.0, 0.9, (T, 2, K_cpu))
    u0 = nominal_warm(T)
    S = orc.savgol_matrix(T)
    out = {}
    for nthr in sorted({1, cores}):
        used = orc.set_threads(nthr)
        orc.get_path([0, 0, 0], goal, u0, eps[:, :, :2000], 0.001, 0.9, S=S)  # warm
        n, t0 = 0, time.perf_counter()
        while True:
            orc.get_path([0, 0, 0], goal, u0, eps, 0.001, 0.9, S=S)
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s / 4 or n >= 20:
                break
        out[used] = K_cpu * n / el
    best = max(out, key=lambda k: out[k])
    # configs 1 and 2, closed loop on the model, >= 20 warm-up and up to 200 timed ticks (time-capped)
    md = {}
    S50 = orc.savgol_matrix(50)
    for name, K in (("c1_K1000", 1000), ("c2_K10000", 10000)):
        noise = np.random.RandomState(0).normal(0.0, 0.9, (50, 2, K))
        for nom_name, nom in (("zero", np.zeros((2, 50))), ("warm", nominal_warm(50))):
            for nthr in sorted({1, cores}):
                used = orc.set_threads(nthr)
                st, lat, ts = np.zeros(3), nom.copy(), []
                cap = time.perf_counter() + budget_s / 16
                for i in range(220):
                    t0 = time.perf_counter()
                    st, _, lat = orc.get_path(st, [0.0, -1.0, 0.0], lat, noise, 0.001, 0.9, S=S50)
                    if i >= 20:
                        ts.append(time.perf_counter() - t0)
                    if time.perf_counter() > cap and len(ts) >= 10:
                        break
                ts = np.sort(np.array(ts))
                md["%s_%s_%dthr" % (name, nom_name, used)] = {
                    "median_ms": float(1e3 * np.median(ts)), "p99_ms": float(1e3 * ts[min(len(ts) - 1, int(0.99 * len(ts)))]),
                    "rollouts_per_s": float(K / np.median(ts)), "ticks": len(ts)}
    orc.set_threads(cores)
    return {"value": out[best], "unit": "rollouts/s", "cores": best, "kind": "port",
            "by_threads": {str(k): v for k, v in out.items()}, "usable_cores": cores,
            "sample": "oracle get_path ticks (fp64, injected noise, OpenMP over K), K=%d T=%d, same goal/nominal as the "
                      "GPU workload; noise generation excluded" % (K_cpu, T),
            "baseline_md_inputs": md}


def compact_line(line, full_path):
    """The printed line: the contract's keys; `roofline` and `cpu_baseline` flattened to scalars -- every figure a reader needs to redo the
    arithmetic is a top-level member of its object:
      roofline.frac                  = VALU issue cycles of the dominant kernel's launch / (avg_launch_us x clock_mhz_under_load)   [the roof it is ON]
      roofline.frac_at_peak_clock    = the same at the 2.4 GHz peak clock
      roofline.hbm_rollout_frac      = hbm_rollout_bytes (PMC counters, = `traffic`) / avg_launch_us / 8 TB/s
      roofline.hbm_update_frac       = hbm_update_bytes / hbm_update_us / 8 TB/s   (hbm_update_frac_of_achievable: against 6.3 TB/s)
      roofline.accounting_8d_frac    = SURVEY 8(d): 12 B x state-steps of the launch / avg_launch_us / 8 TB/s (an accounting figure: eps is never stored)
      roofline.tick_accounting_8d_frac = 24 B x state-steps of the tick / tick time / 8 TB/s (can exceed 1 for the same reason)
      roofline.tick_floor_us, tick_frac = max(rollout issue time, update bytes / 6.3 TB/s) + measured merge + finalize; over the tick
    plus the one-engine and all-fp64 legs as scalars.  The full nested record is in `full_record`."""
    roof, cpu = line.get("roofline") or {}, line.get("cpu_baseline")
    hbm = roof.get("hbm") or {}
    hr, hu = hbm.get("rollout") or {}, hbm.get("update") or {}
    acc, valu = roof.get("accounting_8d") or {}, roof.get("valu") or {}
    tl = (roof.get("tick_level") or {})
    one, f64 = line.get("one_engine") or {}, line.get("f64_storage") or {}
    r = {"kernel": roof.get("kernel"), "bound": roof.get("bound"), "achieved": roof.get("achieved"), "peak": roof.get("peak"), "unit": roof.get("unit"),
         "frac": roof.get("frac"), "traffic": roof.get("traffic"), "frac_at_peak_clock": roof.get("frac_at_peak_clock"),
         "avg_launch_us": roof.get("avg_launch_us"), "launches_timed": roof.get("launches_timed"), "samples_per_launch": roof.get("samples_per_launch"),
         "clock_mhz_under_load": roof.get("clock_mhz_under_load"), "min_launch_us": roof.get("min_launch_us"),
         "valu_per_step": valu.get("valu_per_step"), "issue_cycles_per_step": valu.get("issue_cycles_per_step"),
         "hbm_peak_gbs": HBM_PEAK_GBS, "hbm_achievable_gbs": HBM_ACHIEVABLE_GBS,
         "hbm_rollout_bytes": hr.get("counter_bytes"), "hbm_rollout_gbs": hr.get("achieved"), "hbm_rollout_frac": hr.get("frac"),
         "hbm_update_bytes": hu.get("counter_bytes"), "hbm_update_us": hu.get("avg_launch_us"), "hbm_update_gbs": hu.get("achieved"),
         "hbm_update_frac": hu.get("frac"), "hbm_update_frac_of_achievable": hu.get("frac_of_achievable"),
         "accounting_8d_bytes": acc.get("algorithmic_bytes_per_launch"), "accounting_8d_gbs": acc.get("achieved"), "accounting_8d_frac": acc.get("frac"),
         "tick_accounting_8d_bytes": (tl.get("accounting_8d") or {}).get("algorithmic_bytes"), "tick_accounting_8d_frac": (tl.get("accounting_8d") or {}).get("frac"),
         "tick_valu_frac": tl.get("valu_frac"), "tick_floor_us": roof.get("tick_floor_us"), "tick_frac": roof.get("tick_frac"),
         "measured_in": "one_engine leg (co_shards = 1) of this command" if roof.get("measured_in") else "the timed region of this command",
         "one_engine_ms": one.get("ms_per_step"), "one_engine_rollout_us": one.get("rollout_us"),
         "f64_ms": f64.get("ms_per_step"), "f64_rollout_us": f64.get("rollout_us"), "f64_update_us": f64.get("update_us"),
         "f64_kernel": f64.get("rollout_kernel"), "f64_parked_ms": (f64.get("parked_at_goal") or {}).get("ms_per_step"),
         "f64_valu_issue_frac": f64.get("valu_issue_frac")}
    co = roof.get("co_scheduled_launch") or {}
    if co:
        r.update({"co_launch_samples": co.get("samples_per_launch"), "co_launch_us": co.get("avg_launch_us"), "co_launches": co.get("concurrent_launches")})
    c = None
    if cpu:
        c = {k: cpu.get(k) for k in ("value", "unit", "cores", "kind", "usable_cores")}
        c["sample"] = (cpu.get("sample") or "")[:200]
        for thr, v in (cpu.get("by_threads") or {}).items():
            c["threads_%s_value" % thr] = v
        # the reference's OWN Python loop (control/src/mppi:85-102 imported with ROS stubbed), 1 core, K = 1000 / 10 000, T = 50: quoted from
        # BASELINE.md section 2 (measured in the survey container; the reference does not travel to the GPU box), never re-measured here
        c["python_reference_value"] = PYTHON_REFERENCE_ROLLOUTS_PER_S
        c["python_reference_source"] = "BASELINE.md 2 (quoted: reference's numpy loop, 1 core, K=1000 T=50)"
    cfg = {k: line["config"].get(k) for k in ("workload", "agents", "samples_total", "horizon", "samples_per_gpu", "state_steps_per_tick", "storage", "noise",
                                              "parallelism", "graph", "tick_kernels", "co_shards", "co_samples", "kernels_pinned_by_samples_total")}
    sync, tick = line.get("sync_tick_us") or {}, line.get("tick_us") or {}
    out = {k: line.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                    "dtype", "data")}
    out.update({"config": cfg, "roofline": r, "cpu_baseline": c,
                "tick_us_median": tick.get("median"), "tick_us_p99": tick.get("p99"),
                # the node's own call: fresh state in, blocking, controls out (mppi_tick) -- what Controller.pos_cb pays per odometry message
                "sync_tick_us_median": sync.get("median"), "sync_tick_us_p99": sync.get("p99"),
                "kernels_us": {k: (round(v, 2) if v else v) for k, v in (line.get("kernels_us") or {}).items()},
                "exchange_us": line.get("exchange_us"), "value_parked_at_goal": line.get("value_parked_at_goal"),
                # N > 1: what every rank ran (its share, its big kernels, the exchange it came up on)
                "per_rank": [{"rank": r.get("rank"), "samples": r.get("samples"), "rollout_us": (r.get("kernels_us") or {}).get("rollout"),
                              "update_us": (r.get("kernels_us") or {}).get("update"), "exchange_us": r.get("exchange_us"),
                              "exchange_ran": (r.get("exchange") or {}).get("ran"), "rccl_ranks": r.get("rccl_ranks")}
                             for r in (line.get("per_rank") or [])] or None,
                "one_engine_ms": one.get("ms_per_step"), "f64_ms": f64.get("ms_per_step"),
                "final_state": line.get("final_state"), "final_u": line.get("final_u"),
                "dtype_detail": (line.get("dtype_detail") or "")[:160], "full_record": full_path})

    def trim(v, keep=False):   # seven significant digits are plenty for a record (`value`, `ms_per_step` and the final state stay as they are)
        if isinstance(v, float) and not keep:
            return float("%.7g" % v)
        if isinstance(v, dict):
            return {k: trim(x, keep or k in ("value", "ms_per_step", "final_state", "final_u")) for k, x in v.items()}
        if isinstance(v, list):
            return [trim(x, keep) for x in v]
        return v
    out = {k: trim(v, k in ("value", "ms_per_step", "final_state", "final_u")) for k, v in out.items()}
    return out


class HipEvents(object):
    """HIP events on an arbitrary stream through ctypes (torch.cuda.Event only sees torch's current stream)."""

    def __init__(self):
        # torch is imported before the first engine exists, so libmppi_hip.so is bound to the runtime torch bundles
        # (same SONAME) and this name resolves to that one runtime as well (sharded._one_hip_runtime checks it)
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        self.hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
        self.hip.hipEventSynchronize.argtypes = [C.c_void_p]
        self.pool = []

    def record(self, stream):
        ev = C.c_void_p()
        assert self.hip.hipEventCreate(C.byref(ev)) == 0
        assert self.hip.hipEventRecord(ev, C.c_void_p(stream)) == 0
        self.pool.append(ev)
        return ev

    def elapsed_us(self, a, b):
        ms = C.c_float()
        assert self.hip.hipEventSynchronize(b) == 0
        assert self.hip.hipEventElapsedTime(C.byref(ms), a, b) == 0
        return 1e3 * ms.value

    def free(self):
        for ev in self.pool:
            self.hip.hipEventDestroy(ev)
        self.pool = []


def dist_stats(us):
    us = np.sort(np.asarray(us, dtype=np.float64))
    return {"mean": float(us.mean()), "median": float(np.median(us)),
            "p99": float(us[min(len(us) - 1, int(0.99 * len(us)))]), "ticks": int(len(us))}


def load_profile(name):
    path = os.path.join(ROOT, "profiles", name)
    return json.load(open(path)) if os.path.exists(path) else None


def run_pentagon(args, K, T, local_rank):
    """Config 3: Controller (control/src/mppi:296-389) from the origin through the pentagon with blocking ticks; a step
    is a callback that runs a control tick (goal-switch callbacks only reset the nominal controls, :356-375).  Warm-up
    = throw-away ticks of the same engine (count AND wall time), then MPPI.initialize() and a fresh Controller: the
    timed region is always the first --steps ticks of the same run (4221 ticks cover all five waypoints)."""
    from motion_planning_amd import MPPI, Controller, rk4
    m = MPPI(horizon=T, samples=K, rng="philox", seed=0, storage=args.storage, device=local_rank)
    st, g = np.zeros(3), np.array([1.0, 0.0, 0.0])
    t_w, n_w = time.perf_counter(), 0
    while n_w < args.warmup or time.perf_counter() - t_w < args.min_warmup_s:
        st = m.get_path(st, g)
        n_w += 1
    m.initialize()
    m._tick = 1_000_000          # the timed run's noise streams do not depend on the warm-up's length
    c = Controller(PENTAGON, mppi=m)
    plant = np.array([0.0, 0.0, 0.0])
    timed, switches = [], 0
    while len(timed) < args.steps:
        tick_before, idx_before = m._tick, c.idx
        t0 = time.perf_counter()
        c.pos_cb(plant[0], plant[1], plant[2])
        dt = time.perf_counter() - t0
        if m._tick > tick_before:
            timed.append(dt)
        switches += int(c.idx != idx_before)
        u = np.array([0.0, 0.0]) if c.done else m.uvec[-1, :].copy()
        plant = rk4(plant, u, m.dt)
    return m._eng, np.array(timed), {"waypoint_switches": switches, "warmup_ticks_run": n_w, "final_pose": [float(x) for x in plant],
                                     "goal": [float(x) for x in m.goal]}


def self_launch(n):
    """`python bench.py --gpus N` started as a PLAIN process (no launcher, WORLD_SIZE unset): this process becomes the launcher --
    N children of this same command line, one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment (what
    torch.distributed.run would set; rendezvous on 127.0.0.1, a free port) -- and rank 0's child prints the one JSON line on the
    stdout it inherits.  A rank that fails takes the others down with it (exact pids, never a pattern); the exit code is the
    first non-zero one."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MPPI_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL, the p2p mailboxes)
        env.setdefault("OMP_NUM_THREADS", "1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        live = list(procs)
        while live:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in live:       # one rank down: the others would wait in a collective until its timeout
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--min-warmup-s", type=float, default=0.35,
                    help="keep warming up until this much wall time has passed as well (clocks ramp by time, not by count)")
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--storage", default="f32", choices=["f32", "f64"])
    ap.add_argument("--samples", type=int, default=0, help="override K_total")
    ap.add_argument("--samples-total", type=int, default=0,
                    help="with --samples K on ONE GPU: measure a rank's SHARE of a controller of this many samples (mppi_config.samples_total: the "
                         "kernels the whole controller's size picks, as every rank of `--gpus N` runs them; no exchange partner)")
    ap.add_argument("--rank-kernels", action="store_true",
                    help="N > 1: let every rank pick its kernels by its OWN share (samples_total = 0: faster small shares, "
                         "results no longer equal across N beyond 1e-6)")
    ap.add_argument("--horizon", type=int, default=0, help="override T (sweeps; not a BASELINE config)")
    ap.add_argument("--tick-path", default="auto", choices=["auto", "lanes", "scan"],
                    help="which kernels a tick runs (include/mppi_hip.h MPPI_TICK_*; sweeps)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "rccl", "p2p"],
                    help="N > 1: how the [A][T][8] partials cross GPUs (auto = p2p over IPC-mapped mailboxes when the probe passes, else RCCL)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f64-line", action="store_true", help="skip the extra all-fp64 measurement (N = 1, c4)")
    ap.add_argument("--graph", action="store_true", help="replay the tick as one hipGraph (N=1 only)")
    ap.add_argument("--no-co-line", action="store_true", help="(kept for old command lines; the co-scheduled tick is the headline handle's own now)")
    ap.add_argument("--co-shards", type=int, default=None, help="mppi_config.co_shards of the measured engine (default: the engine's own rule)")
    ap.add_argument("--full-out", default="", help="where the full nested record goes (default: gpurun_out/ on a gpurun box, else profiles/)")
    ap.add_argument("--group-of-one", action="store_true",
                    help="TEST ONLY: with one rank, still initialise the process group and tick through tick_begin -> all-gather -> "
                         "tick_finish (RCCL as a world of one); the default `--gpus 1` never does, launcher or not")
    ap.add_argument("--all-ranks-on-gpu0", action="store_true",
                    help="TEST ONLY: every rank drives cuda:0 and the process group is gloo (RCCL refuses two ranks on one "
                         "device) -- runs the N > 1 code path of this script on a one-GPU box; use with --exchange p2p")
    args = ap.parse_args()

    # However this script is started, `--gpus N` runs N ranks, one process per GPU:
    #   under a launcher (torch.distributed.run, the driver's N > 1 command line): WORLD_SIZE / RANK / LOCAL_RANK are in the environment;
    #   as a plain process with N > 1: it launches the N ranks itself (self_launch) and returns their exit code.
    # `--gpus 1` is the SAME measurement whether or not a launcher wrapped it: one process, no process group, the handle's own
    # (co-scheduled) fused tick -- a world of one has nothing to exchange.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist
    from motion_planning_amd import sharded

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world        # (a launcher's world size is what runs)
    if args.all_ranks_on_gpu0:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    in_group = world > 1 or (args.group_of_one and "MASTER_PORT" in os.environ)
    coll_dev = "cpu" if args.all_ranks_on_gpu0 else "cuda"       # where the few control collectives of this script live
    if in_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.all_ranks_on_gpu0:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    desc, A_total, K_total, T, goal = WORKLOADS[args.workload]
    if args.samples:
        K_total = args.samples
    if args.horizon:
        T = args.horizon
    extra = {}

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.workload == "c3":
        if world > 1:
            raise SystemExit("config 3 is a single-GPU configuration")
        eng, timed, extra["pentagon"] = run_pentagon(args, K_total, T, local_rank)
        A, K_local, units_total = 1, K_total, K_total
        elapsed = float(timed.sum())
        ms_per_step = 1e3 * elapsed / args.steps
        tick_us = dist_stats(1e6 * timed)
        ktimes = dtimes = {k: (0.0, 0) for k in ("nominal", "rollout", "update", "merge", "finalize", "exchange")}
        btimes, sync_tick_us, exchange_us, exch_kind = None, tick_us, None, "none"
        # one more leg of diagnostic ticks with every kernel bracketed
        eng.kernel_timing(("nominal", "rollout", "update", "merge", "finalize"), period=1)
        st = np.array([extra["pentagon"]["final_pose"]])
        for i in range(20):
            st, _ = eng.tick(st, np.array([[1.0, 0.0, 0.0]]), noise="philox", seed=0, tick_id=20_000_000 + i)
        dtimes = eng.kernel_times()
        ktimes = dtimes
        eng.kernel_timing(())
        final_nxt, final_ua = eng.get_outputs()
        timed_nxt, timed_ua, clock_mhz = final_nxt, final_ua, eng.shader_clock_mhz()
    else:
        if args.workload == "c5":  # independent agents: replicas, no collective
            lo, hi = sharded.shard_range(A_total, world, rank)
            A = hi - lo
            ticker, eng = sharded.make_replica_ticker(K_total, T, n_agents=A, storage=args.storage, local_rank=local_rank,
                                                      tick_path=args.tick_path, co_shards=args.co_shards, agent_offset=lo)
            states = np.array([[0.05 * a, 0.0, 0.0] for a in range(lo, hi)])
            goals = np.array([[0.05 * a, -1.0, 0.0] for a in range(lo, hi)])
            K_local, units_total = K_total, A_total * K_total
        else:
            # every rank (N = 1 included) is told the whole controller's size: all N run the same arithmetic
            pinned = 0 if args.rank_kernels else (args.samples_total or K_total)
            ticker, eng = sharded.make_hip_ticker(K_total, T, n_agents=1, storage=args.storage, local_rank=local_rank,
                                                  exchange=args.exchange, tick_path=args.tick_path, co_shards=args.co_shards,
                                                  pinned_total=pinned)
            extra["samples_total_pinned"] = pinned
            A = 1
            states, goals = np.zeros((1, 3)), np.array([goal])
            K_local, units_total = eng.K, K_total
        exch_kind = ticker.exchange
        for a in range(A):
            eng.set_nominal(nominal_warm(T), agent=a)

        seed = 0
        counter = [0]

        def tick(first=False):
            i = counter[0]
            counter[0] += 1
            if args.graph and not in_group and not first:
                eng.tick_graph(seed)
            else:
                ticker.tick_async(states if first else None, goals if first else None, "philox", seed, i)

        # warm-up: at least --warmup ticks AND at least --min-warmup-s of wall time (a 20-tick warm-up of this
        # workload is 4 ms: the clocks have not ramped yet and the first timed ticks run slow)
        eng.kernel_timing(("rollout", "update"), period=8)   # the warm-up's launches too: rocprofv3 averages over the whole command
        tick(first=True)
        t_w = time.perf_counter()
        n_w = 0
        while True:
            # in rounds of 16 ticks; rank 0's clock decides when the warm-up ends and tells the others, so that every rank
            # runs the same number of ticks and issues its collectives (the exchange, this broadcast) in the same order
            for _ in range(16):
                tick()
            n_w += 16
            torch.cuda.synchronize()
            done = n_w >= args.warmup and time.perf_counter() - t_w >= args.min_warmup_s
            if world > 1:
                t = torch.tensor([1 if done else 0], dtype=torch.int64, device=coll_dev)
                dist.broadcast(t, src=0)
                done = bool(t.item())
            if done:
                break
        extra["warmup_ticks_run"] = n_w
        wtimes = eng.kernel_times()
        # The warm-up above is real ticks of the closed loop, and a time-based one is long enough to park the robot
        # at its goal -- a different workload (1-5 % of a row carry softmax weight there, a handful do on the way).
        # The measured ticks must not depend on how long the clocks took to ramp: put the controller back at the start
        # (state, nominal controls) so that the timed region is always ticks 1..K of the same parallel-park run.
        phase = [0]

        def restart():
            for a in range(A):
                eng.set_nominal(nominal_warm(T), agent=a)
            phase[0] += 1
            counter[0] = 1_000_000 * phase[0]      # tick ids (= noise streams) of a phase do not depend on the warm-up's length
            tick(first=True)
        restart()
        # Timed region: exactly --steps ticks between barrier + synchronize pairs.  The dominant kernel (rollout) is
        # timed live with HIP events that ride on its own launch (hipExtLaunchKernelGGL start / stop events on the
        # engine's stream = the dispatch's begin / end timestamps, the clock rocprofv3 --kernel-trace reads); no
        # marker packets enter the stream, every EVENT_PERIOD-th launch is sampled only to keep the event pool small.
        EVENT_PERIOD = int(os.environ.get("MPPI_EVENT_PERIOD", "4"))
        eng.kernel_timing(("rollout",), period=EVENT_PERIOD)
        sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            tick()
        sync()
        elapsed = time.perf_counter() - t0
        ktimes = eng.kernel_times()
        timed_nxt, timed_ua = eng.get_outputs()          # where the timed region ended (self-check against the one-engine run)
        clock_mhz = eng.shader_clock_mhz()               # a probe wave inside the last rollout launch
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        ms_per_step = 1e3 * elapsed / args.steps

        # Distribution pass (not the headline clock): one event between consecutive ticks on the engine's stream
        # gives every tick's own duration -> median and p99 of the back-to-back tick.
        eng.kernel_timing(())
        evs = HipEvents()
        stream = eng.get_stream()
        restart()
        sync()
        marks = [evs.record(stream)]
        for i in range(args.steps):
            tick()
            marks.append(evs.record(stream))
        sync()
        tick_us = dist_stats([evs.elapsed_us(marks[i], marks[i + 1]) for i in range(args.steps)])
        evs.free()

        # Diagnostic pass: every kernel bracketed on every launch; N > 1: the exchange bracketed as well.
        n_diag = min(args.steps, 20)
        restart()
        eng.kernel_timing(("nominal", "rollout", "update", "merge", "finalize", "exchange"), period=1)
        ticker.time_exchange(True)
        sync()
        for i in range(n_diag):
            tick()
        sync()
        dtimes = eng.kernel_times()
        exchange_us = ticker.exchange_times_us()
        if exchange_us is None and dtimes["exchange"][1]:   # p2p: the publish kernel; the wait for the peers sits in `finalize`
            exchange_us = dtimes["exchange"][0] * 1e3 / dtimes["exchange"][1]
        ticker.time_exchange(False)
        eng.kernel_timing(())
        nxt, ua = eng.get_outputs()
        assert np.isfinite(nxt).all() and np.isfinite(ua).all()
        final_nxt, final_ua = nxt.copy(), ua.copy()

        # Diagnostic: the other regime of the closed loop -- the robot parked AT its goal, zero nominal controls: most of
        # a row's samples cost about the same, 1-5 % of them carry softmax weight (the update kernel re-draws their noise)
        if not in_group and args.workload == "c4":
            for a in range(A):
                eng.set_nominal(np.zeros((2, T)), agent=a)
            ticker.tick_async(goals, goals, "philox", seed, 5_000_000)
            for i in range(30):
                ticker.tick_async(None, None, "philox", seed, 5_000_001 + i)
            sync()
            t0 = time.perf_counter()
            for i in range(50):
                ticker.tick_async(None, None, "philox", seed, 5_000_100 + i)
            sync()
            el = time.perf_counter() - t0
            eng.kernel_timing(("rollout", "update", "merge", "finalize"), period=1)   # a second pass for the kernel times: brackets cost stream time
            for i in range(20):
                ticker.tick_async(None, None, "philox", seed, 5_000_200 + i)
            sync()
            pt = eng.kernel_times()
            eng.kernel_timing(())
            extra["parked_at_goal"] = {"ms_per_step": 1e3 * el / 50, "value": units_total / (el / 50),
                                       "kernels_us": {k: (v[0] * 1e3 / v[1] if v[1] else None) for k, v in pt.items()}}
            for a in range(A):
                eng.set_nominal(nominal_warm(T), agent=a)
            nxt, _ = eng.tick(states, goals, noise="philox", seed=seed, tick_id=6_000_000)

        # Diagnostic: the node's own call pattern -- host state in, blocking, host controls out
        # (mppi_tick; what Controller.pos_cb pays per odometry message), N = 1 only.
        sync_tick_us, btimes = None, None
        if not in_group:
            n_lat = 200     # (whatever --steps says: twenty calls are no statistic, and the first ones behind a change of call pattern run slow)
            # NOTHING bracketed while the call is timed (the event pair around every rollout launch cost the blocking call
            # 30 us at config 4); the rollout launch of this call pattern is timed in a short pass of its own behind it
            eng.kernel_timing(())
            st, lat = nxt, []
            # the interpreter's cyclic garbage collector is held off for these calls: with torch and numpy imported a full
            # collection takes 30-45 ms and lands on a fixed call of this loop (call 11 at --steps 200, call 24 at 100:
            # allocation counts, not the engine -- rounds 2 and 3 reported it as `max`); a node written in Python does the same
            gc_was_on = gc.isenabled()
            gc.collect()
            gc.disable()
            for i in range(20):     # (the call pattern's own warm-up, not recorded)
                st, _ = eng.tick(st, goals, noise="philox", seed=seed, tick_id=9_000_000 + i)
            for i in range(n_lat):
                t0 = time.perf_counter()
                st, _ = eng.tick(st, goals, noise="philox", seed=seed, tick_id=10_000_000 + i)
                lat.append(1e6 * (time.perf_counter() - t0))
            eng.kernel_timing(("rollout",), period=1)
            for i in range(min(n_lat, 20)):
                st, _ = eng.tick(st, goals, noise="philox", seed=seed, tick_id=11_000_000 + i)
            if gc_was_on:
                gc.enable()
            btimes = eng.kernel_times()
            eng.kernel_timing(())
            sync_tick_us = dist_stats(lat)
            sync_tick_us["instrumented"] = False
            sync_tick_us["max"] = float(max(lat))
            sync_tick_us["max_at_call"] = int(np.argmax(lat))
            sync_tick_us["first_calls"] = [float(x) for x in lat[:4]]

    info = eng.info()
    # per-rank kernel times (N > 1): gathered on rank 0
    kernels_us = {name: (dtimes[name][0] * 1e3 / dtimes[name][1] if dtimes[name][1] else None) for name in dtimes}
    per_rank = None
    if world > 1:
        objs = [None] * world
        dist.all_gather_object(objs, {"rank": rank, "device": local_rank, "kernels_us": kernels_us, "exchange_us": exchange_us,
                                      "samples": int(eng.K), "exchange": getattr(ticker, "exchange_report", None),
                                      # ranks of the RCCL communicator this rank is in (0: the group is gloo -- the one-GPU test mode)
                                      "rccl_ranks": world if dist.get_backend() == "nccl" else 0,
                                      "shader_clock_mhz": clock_mhz, "co_shards": info.get("co_shards", 1)})
        per_rank = objs

    # All-fp64 line next to the fp32-storage one (the reference is float64 end to end): N = 1, config 4 only.
    f64_line = None
    if (rank == 0 and world == 1 and args.workload == "c4" and args.storage == "f32" and not args.no_f64_line
            and not args.samples):
        eng.close()
        t64, e64 = sharded.make_hip_ticker(K_total, T, n_agents=1, storage="f64", local_rank=local_rank)
        e64.set_nominal(nominal_warm(T))
        # the headline's own protocol: its own engine, a warm-up by count AND wall time, the controller put back at the
        # start (a time-based warm-up parks the robot at its goal: another workload), then --steps timed ticks of the same tick ids
        t64.tick_async(np.zeros((1, 3)), np.array([goal]), "philox", 0, 0)
        t_w, i = time.perf_counter(), 1
        while time.perf_counter() - t_w < args.min_warmup_s or i < args.warmup:
            t64.tick_async(None, None, "philox", 0, i)
            i += 1
            if i % 16 == 0:
                e64.synchronize()
        e64.set_nominal(nominal_warm(T))
        t64.tick_async(np.zeros((1, 3)), np.array([goal]), "philox", 0, 1_000_000)
        e64.kernel_timing(("rollout", "update"), period=4)
        e64.synchronize()
        n64 = max(args.steps, 100)     # (an auxiliary leg: twenty ticks right behind the fp32 legs are 3.6 ms of a chip still settling -- its `steps` field says what was timed)
        t0 = time.perf_counter()
        for j in range(n64):
            t64.tick_async(None, None, "philox", 0, 1_000_001 + j)
        e64.synchronize()
        el64 = time.perf_counter() - t0
        k64 = e64.kernel_times()
        kind64 = e64.info().get("rollout_kernel")
        mhz64 = e64.shader_clock_mhz()
        # the other regime: parked at the goal (the engine goes back to the two-kernel tick there: rollout + update)
        e64.kernel_timing(())
        e64.set_nominal(np.zeros((2, T)))
        t64.tick_async(np.array([goal]), np.array([goal]), "philox", 0, 5_000_000)
        for j in range(30):
            t64.tick_async(None, None, "philox", 0, 5_000_001 + j)
        e64.synchronize()
        t0 = time.perf_counter()
        for j in range(50):
            t64.tick_async(None, None, "philox", 0, 5_000_100 + j)
        e64.synchronize()
        elp64 = time.perf_counter() - t0
        f64_line = {"storage": "f64", "dtype": "f64", "ms_per_step": 1e3 * el64 / n64, "value": K_total / (el64 / n64), "steps": n64,
                    "protocol": "as the headline: own engine, warm-up, controller back at the start, --steps timed ticks",
                    "rollout_kernel": kind64,
                    "rollout_us": k64["rollout"][0] * 1e3 / max(k64["rollout"][1], 1),
                    "update_us": k64["update"][0] * 1e3 / max(k64["update"][1], 1) if k64["update"][1] else None,
                    "shader_clock_mhz": mhz64,
                    "parked_at_goal": {"ms_per_step": 1e3 * elp64 / 50, "rollout_kernel": e64.info().get("rollout_kernel")},
                    "note": "V and the softmax in fp64: the reference's own precision end to end.  `fused`: rollout + cost-to-go + softmax "
                            "partials in ONE kernel, V never stored (under way); parked at the goal the engine runs rollout + update"}
        e64.close()

    # One engine next to the headline: N = 1, config 4 only.  The headline handle splits its fused tick over co-scheduled
    # engines (mppi_config.co_shards AUTO: one shard's HBM-bound update kernel runs under the other's VALU-bound rollout); with
    # kernels of several engines overlapping, a launch's own duration is no longer a kernel's undisturbed time, so the
    # per-kernel rooflines (SURVEY 8d accounting, VALU issue) are measured on the SAME workload with co_shards = 1 -- same
    # protocol (time-based warm-up, controller put back at the start, the same tick ids timed).  It is also the self-check of
    # the timed region: the two runs must end in the same state and controls (a timed region that skipped work would not).
    one_line = None
    if rank == 0 and world == 1 and args.workload == "c4" and not args.samples and not args.graph and info.get("co_shards", 1) > 1:
        if f64_line is None:
            eng.close()
        from motion_planning_amd.mppi import Engine
        with Engine(K_total, T, storage=args.storage, device=local_rank, tick_path=args.tick_path, co_shards=1) as e1:
            e1.set_nominal(nominal_warm(T))
            e1.tick_async(np.zeros((1, 3)), np.array([goal]), "philox", 0, 0)
            t_w, i = time.perf_counter(), 1
            while time.perf_counter() - t_w < args.min_warmup_s or i < args.warmup:
                e1.tick_async(None, None, "philox", 0, i)
                i += 1
                if i % 16 == 0:
                    e1.synchronize()
            e1.set_nominal(nominal_warm(T))
            e1.tick_async(np.zeros((1, 3)), np.array([goal]), "philox", 0, 1_000_000)
            e1.kernel_timing(("rollout",), period=4)
            e1.synchronize()
            t0 = time.perf_counter()
            for j in range(args.steps):
                e1.tick_async(None, None, "philox", 0, 1_000_001 + j)
            e1.synchronize()
            el1 = time.perf_counter() - t0
            k1 = e1.kernel_times()
            mhz1 = e1.shader_clock_mhz()
            kind1 = e1.info().get("rollout_kernel")
            nxt1, ua1 = e1.get_outputs()
            # every kernel bracketed
            e1.kernel_timing(("rollout", "update", "merge", "finalize"), period=1)
            for j in range(20):
                e1.tick_async(None, None, "philox", 0, 3_000_000 + j)
            e1.synchronize()
            d1 = e1.kernel_times()
            # the other regime: parked at the goal (1-7 % of a row carry weight, the update kernel re-draws their noise)
            e1.set_nominal(np.zeros((2, T)))
            e1.tick_async(np.array([goal]), np.array([goal]), "philox", 0, 5_000_000)
            for j in range(30):
                e1.tick_async(None, None, "philox", 0, 5_000_001 + j)
            e1.synchronize()
            t0 = time.perf_counter()
            for j in range(50):
                e1.tick_async(None, None, "philox", 0, 5_000_100 + j)
            e1.synchronize()
            elp = time.perf_counter() - t0
            e1.kernel_timing(("rollout", "update", "merge", "finalize"), period=1)
            for j in range(20):
                e1.tick_async(None, None, "philox", 0, 5_000_200 + j)
            e1.synchronize()
            p1 = e1.kernel_times()
            e1.kernel_timing(())
        one_line = {"co_shards": 1, "ms_per_step": 1e3 * el1 / args.steps, "value": K_total / (el1 / args.steps), "steps": args.steps,
                    "rollout_us": k1["rollout"][0] * 1e3 / max(k1["rollout"][1], 1), "launches_timed": k1["rollout"][1],
                    "shader_clock_mhz": mhz1, "rollout_kernel": kind1,
                    "kernels_us_bracketed": {k: (v[0] * 1e3 / v[1] if v[1] else None) for k, v in d1.items()},
                    "final_state": [float(x) for x in nxt1[0]], "final_u": [float(x) for x in ua1[0]],
                    "parked_at_goal": {"ms_per_step": 1e3 * elp / 50, "value": K_total / (elp / 50),
                                       "kernels_us": {k: (v[0] * 1e3 / v[1] if v[1] else None) for k, v in p1.items()}}}
        dev_s = float(np.abs(np.array(one_line["final_state"]) - timed_nxt[0]).max())
        dev_u = float(np.abs(np.array(one_line["final_u"]) - timed_ua[0]).max())
        one_line["self_check"] = {"max_abs_diff_state": dev_s, "max_abs_diff_u": dev_u, "tolerance": 1e-10,
                                  "what": "final state / applied controls of the timed region: co-scheduled headline vs one engine, same seed, same ticks"}
        if not (dev_s <= 1e-10 and dev_u <= 1e-10):
            raise SystemExit("bench self-check FAILED: the timed region of the headline run and of the one-engine run ended in "
                             "different states / controls: %r" % one_line["self_check"])

    # The optional 16-bit noise packing (option "noise_packing" = 1: one Philox call serves four steps instead of three) next to
    # the default stream: N = 1, config 4, fp32 storage only; the handle's own co-scheduling, then one engine with its rollout timed.
    pack_line = None
    if (rank == 0 and world == 1 and args.workload == "c4" and args.storage == "f32" and not args.samples and not args.graph
            and not args.no_f64_line and one_line is not None):
        from motion_planning_amd.mppi import Engine
        pack_line = {"option": "noise_packing = 1 (not the default: Box-Muller radius cut at 4.85 sigma instead of 5.53, 2^16 directions)"}
        for name, co in (("co_scheduled", None), ("one_engine", 1)):
            with Engine(K_total, T, storage="f32", device=local_rank, tick_path=args.tick_path, co_shards=co,
                        options={"noise_packing": 1}) as ep:
                ep.set_nominal(nominal_warm(T))
                ep.tick_async(np.zeros((1, 3)), np.array([goal]), "philox", 0, 0)
                t_w, i = time.perf_counter(), 1
                while time.perf_counter() - t_w < args.min_warmup_s or i < args.warmup:   # (the headline's own protocol)
                    ep.tick_async(None, None, "philox", 0, i)
                    i += 1
                    if i % 16 == 0:
                        ep.synchronize()
                ep.set_nominal(nominal_warm(T))
                ep.tick_async(np.zeros((1, 3)), np.array([goal]), "philox", 0, 1_000_000)
                if co == 1:
                    ep.kernel_timing(("rollout",), period=4)
                ep.synchronize()
                t0 = time.perf_counter()
                for j in range(args.steps):
                    ep.tick_async(None, None, "philox", 0, 1_000_001 + j)
                ep.synchronize()
                elq = time.perf_counter() - t0
                pack_line[name] = {"ms_per_step": 1e3 * elq / args.steps, "value": K_total / (elq / args.steps), "steps": args.steps}
                if co == 1:
                    kq = ep.kernel_times()
                    pack_line[name]["rollout_us"] = kq["rollout"][0] * 1e3 / max(kq["rollout"][1], 1)
                    ep.kernel_timing(())

    if rank == 0:
        lanes = info.get("tick_kernels", "lanes") == "lanes"
        mixes = load_profile(PROFILE_ROUND + "_valu_mix.json") or load_profile("r5_valu_mix.json") or load_profile("r4_valu_mix.json") or {}

        pmc_name = PROFILE_ROUND + "_pmc_summary_bench_c4.json"
        pm = load_profile(pmc_name)
        for older in ("r5", "r4", "r3", "r1"):
            if pm is None:
                pmc_name = older + "_pmc_summary_bench_c4.json"
                pm = load_profile(pmc_name)
        pmc_ok = bool(pm) and args.workload == "c4" and args.storage == "f32" and not args.samples and args.horizon in (0, 50)

        def pmc_of(kernel_name):
            """Per-launch PMC averages of one kernel over ALL of config 4's samples (tools/pmc.sh: separate rocprofv3 --pmc passes of
            `bench.py --co-shards 1`; FETCH_SIZE doubled -- the guide's gfx950 correction --, WRITE_SIZE as reported), or None."""
            if not pmc_ok:
                return None
            for kname, c in pm.items():
                if kname.split("::")[-1].split("<")[0] == kernel_name:
                    rd = [v for k, v in c.items() if k.startswith("hbm_read_bytes")]
                    wr = [v for k, v in c.items() if k.startswith("hbm_write_bytes")]
                    if rd and wr:
                        return {"read": rd[0], "write": wr[0], "valu_insts": c.get("SQ_INSTS_VALU")}
            return None

        def kernel_roofline(k_launch, avg_s, mhz, n_timed, kind, upd_us=None, merge_us=None, fin_us=None, tick_s=None, a_launch=None):
            """The dominant kernel of a launch over k_launch samples per agent.
            TOP LEVEL = the roof it is on: VALU issue -- wave-instructions of the steady-state loop from the compiler's own assembly
            (tools/valu_mix.py), each class at its measured issue cost (tools/ubench.hip, shader clock read in-kernel), over 1024
            SIMDs at the clock a probe wave inside the launch measured (`frac`; `frac_at_peak_clock` at 2.4 GHz).
            `hbm`: what the kernels really move (PMC counters) against the 8 TB/s peak -- rollout and update.
            `accounting_8d`: SURVEY 8(d)'s figure -- 12 algorithmic B / state-step / kernel over the launch duration -- kept as the
            contract defines it; it is an accounting figure, not a distance to a roof.
            `tick_floor_us`: max(rollout issue time, update bytes / achievable HBM rate) + measured merge + finalize."""
            A_l = a_launch or A              # agents one launch covers (a handle that splits its AGENTS over two engines: half of them)
            steps = A_l * k_launch * T
            share = (A_l * k_launch) / float(A_total * K_total)    # the PMC passes covered ALL samples of the workload: a shard's (or a rank's) launch moves its share
            # `kind`: what the engine says its last tick launched (mppi_rollout_kernel), not a copy of its rule
            name = {"mixed": "rollout_pk_kernel", "fp64": "rollout_kernel", "scan": "scan_tick_kernel", "fused": "rollout_fused_kernel"}[kind]
            gbs = BYTES_PER_STEP_PER_KERNEL * steps / avg_s / 1e9
            acc = {"bound": "hbm", "bytes_per_state_step_per_kernel": BYTES_PER_STEP_PER_KERNEL, "algorithmic_bytes_per_launch": BYTES_PER_STEP_PER_KERNEL * steps,
                   "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                   "note": "SURVEY 8(d): eps 2 x fp32 + V fp32 per state-step, written once by the rollout and read once by the update = 12 B/step per "
                           "kernel, 24 per tick.  NOT the distance to a roof: eps is never stored (it is re-drawn from its Philox counter), so the "
                           "kernels move about a third of these bytes (`hbm`), and the tick-level figure can exceed 1"}
            r = {"kernel": name, "bound": "valu-issue", "achieved": None, "peak": None, "unit": "G wave-inst/s", "frac": None, "traffic": None,
                 "samples_per_launch": A_l * k_launch, "avg_launch_us": avg_s * 1e6, "launches_timed": n_timed, "accounting_8d": acc}
            mix = mixes.get(name)
            if mix and lanes and (args.storage == "f32" or name == "rollout_fused_kernel"):   # (the fused fp64 kernel: its rollout loop's issue time; the fold behind each group is not in it)
                wave_steps = steps / 64.0          # sample-steps per 64 lanes (the pk kernel's waves carry 128 samples)
                cyc = mix["issue_cycles_per_step"] * wave_steps / N_SIMD
                clk = mhz * 1e6 if mhz and mhz > 0 else CLOCK_PEAK_HZ
                r.update({"achieved": mix["valu_per_step"] * wave_steps / avg_s / 1e9, "peak": N_SIMD * clk / mix["avg_cycles_per_valu"] / 1e9,
                          "frac": cyc / clk / avg_s, "frac_at_peak_clock": cyc / CLOCK_PEAK_HZ / avg_s, "clock_mhz_under_load": mhz,
                          "min_launch_us": cyc / clk * 1e6, "min_launch_us_at_peak_clock": cyc / CLOCK_PEAK_HZ * 1e6})
                r["valu"] = {"insts_per_launch": mix["valu_per_step"] * wave_steps, "valu_per_step": mix["valu_per_step"],
                             "issue_cycles_per_step": mix["issue_cycles_per_step"], "avg_cycles_per_valu": mix["avg_cycles_per_valu"],
                             "by_class_per_step": {k: v / mix["steps_per_iteration"] for k, v in mix["by_class_per_iteration"].items()},
                             "source": "profiles/%s_valu_mix.json, profiles/%s_ubench.txt" % (PROFILE_ROUND, PROFILE_ROUND)}
            else:   # no assembly count for this kernel / mode: the contract's HBM accounting is what there is
                r.update({"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS})
            hbm = {"peak": HBM_PEAK_GBS, "achievable": HBM_ACHIEVABLE_GBS, "unit": "GB/s",
                   "source": "profiles/%s (FETCH_SIZE x 2 + WRITE_SIZE per launch over all samples of config 4, separate --pmc passes of "
                             "`bench.py --co-shards 1`), scaled to this launch's samples; durations measured live in this run" % pmc_name}
            pr, pu = pmc_of(name), pmc_of("update_kernel")
            if pr:
                b = (pr["read"] + pr["write"]) * share
                r["traffic"] = b
                hbm["rollout"] = {"counter_bytes": b, "read_bytes": pr["read"] * share, "write_bytes": pr["write"] * share, "avg_launch_us": avg_s * 1e6,
                                  "achieved": b / avg_s / 1e9, "frac": b / avg_s / 1e9 / HBM_PEAK_GBS,
                                  "vs_algorithmic": b / (BYTES_PER_STEP_PER_KERNEL * steps)}
                if pr["valu_insts"] and "valu" in r:
                    r["valu"]["insts_per_launch_pmc_all_samples"] = pr["valu_insts"]
            if pu and upd_us:
                b = (pu["read"] + pu["write"]) * share
                hbm["update"] = {"kernel": "update_kernel", "counter_bytes": b, "avg_launch_us": upd_us, "achieved": b / (upd_us * 1e-6) / 1e9,
                                 "frac": b / (upd_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "frac_of_achievable": b / (upd_us * 1e-6) / 1e9 / HBM_ACHIEVABLE_GBS,
                                 "vs_algorithmic": b / (BYTES_PER_STEP_PER_KERNEL * steps)}
            r["hbm"] = hbm
            if "min_launch_us" in r and upd_us and fin_us is not None and tick_s:
                upd_floor = (hbm["update"]["counter_bytes"] / (HBM_ACHIEVABLE_GBS * 1e9) * 1e6) if "update" in hbm else upd_us
                floor = max(r["min_launch_us"], upd_floor) + (merge_us or 0.0) + fin_us
                r["tick_floor_us"] = floor
                r["tick_frac"] = floor / (tick_s * 1e6)
                r["tick_floor_terms"] = {"rollout_issue_us": r["min_launch_us"], "update_bytes_over_achievable_hbm_us": upd_floor,
                                         "merge_us_measured": merge_us or 0.0, "finalize_us_measured": fin_us, "tick_us": tick_s * 1e6,
                                         "what": "max(rollout VALU-issue time at the measured clock, update counter bytes / 6.3 TB/s) + the measured merge "
                                                 "and finalize launches: the two big kernels fully overlapped, the small ones as they are"}
            return r

        if f64_line and mixes.get("rollout_fused_kernel") and f64_line.get("rollout_kernel") == "fused" and f64_line.get("shader_clock_mhz"):
            # the fused fp64 kernel against ITS roof: VALU issue of its rollout loop (the fold behind every group of 64 samples is not counted)
            cyc64 = mixes["rollout_fused_kernel"]["issue_cycles_per_step"] * (K_total * T / 64.0) / N_SIMD
            f64_line["valu_issue_frac"] = cyc64 / (f64_line["shader_clock_mhz"] * 1e6) / (f64_line["rollout_us"] * 1e-6)
            f64_line["valu_issue_frac_at_peak_clock"] = cyc64 / CLOCK_PEAK_HZ / (f64_line["rollout_us"] * 1e-6)
        co_n = info.get("co_shards", 1)
        k_launch = info["co_samples"][0] if co_n > 1 else K_local    # the launches this handle's events time: its own shard
        # co_samples all equal to K: the handle splits its AGENTS (config 5), engine 0 carries the first ceil(A / 2) of them
        agent_split = co_n > 1 and A > 1 and all(k == K_local for k in info["co_samples"])
        a_launch = (A + 1) // 2 if agent_split else None
        ms, n = ktimes["rollout"]
        if n == 0:  # hipGraph replay: launches are not individually bracketed
            ms, n = dtimes["rollout"] if dtimes["rollout"][1] else (float("nan"), 1)
        avg_s = ms * 1e-3 / max(n, 1)
        tick_s = elapsed / args.steps
        roofline = kernel_roofline(k_launch, avg_s, clock_mhz, n, info["rollout_kernel"], upd_us=kernels_us.get("update"),
                                   merge_us=kernels_us.get("merge"), fin_us=kernels_us.get("finalize"), tick_s=tick_s, a_launch=a_launch)
        tick_bytes = 2 * BYTES_PER_STEP_PER_KERNEL * A * K_local * T
        if co_n > 1:
            roofline["concurrent_launches"] = co_n
            roofline["concurrency_note"] = ("this handle runs its fused tick as %d co-scheduled engines: each shard's rollout launch covers %d of the %d samples and "
                                            "runs NEXT TO the other shards' launches (own streams), so its own duration says little about the kernel; the "
                                            "per-kernel figures of this line are those of the one-engine leg" % (co_n, (a_launch or A) * k_launch, A * K_local))
        tick_level = {"accounting_8d": {"algorithmic_bytes": tick_bytes, "achieved": tick_bytes / tick_s / 1e9, "frac": tick_bytes / tick_s / 1e9 / HBM_PEAK_GBS,
                                        "note": "24 B/state-step x all samples of the tick / tick time: both kernels (and, co-scheduled, both engines) together; above 1 "
                                                "means what the note of accounting_8d says -- most of these bytes never exist"}}
        if roofline.get("issue_cycles_per_step") or "valu" in roofline:   # VALU issue over the whole tick: every shard's rollout instructions against the tick time
            clk = (clock_mhz * 1e6) if clock_mhz and clock_mhz > 0 else CLOCK_PEAK_HZ
            cyc_tick = roofline["valu"]["issue_cycles_per_step"] * (A * K_local * T / 64.0) / N_SIMD
            tick_level["valu_frac"] = cyc_tick / clk / tick_s
            tick_level["valu_note"] = "rollout issue cycles of all shards / (tick time x measured clock): the update, merge and finalize kernels' instructions not counted"
        if one_line:
            kb = one_line["kernels_us_bracketed"]
            one_line["roofline"] = kernel_roofline(K_local, one_line["rollout_us"] * 1e-6, one_line["shader_clock_mhz"], one_line["launches_timed"],
                                                   one_line["rollout_kernel"], upd_us=kb.get("update"), merge_us=kb.get("merge"), fin_us=kb.get("finalize"),
                                                   tick_s=one_line["ms_per_step"] * 1e-3)
        # the rollout launch in each phase of this command (what a rocprofv3 --kernel-trace --stats of the whole
        # command averages over): back-to-back ticks run a few % longer than launches behind an idle gap
        phases = {"timed": ktimes["rollout"], "diagnostic": dtimes["rollout"]}
        extra_roof = {}
        if args.workload != "c3":
            phases["warmup (closed loop, parks at the goal after ~600 ticks; sampled 1 in 8)"] = wtimes["rollout"]
            extra_roof["update_us_warmup_phase"] = wtimes["update"][0] * 1e3 / wtimes["update"][1] if wtimes["update"][1] else None
        if btimes:
            phases["blocking"] = btimes["rollout"]
        extra_roof["rollout_us_by_phase"] = {k: {"avg_us": (v[0] * 1e3 / v[1] if v[1] else None), "launches_timed": v[1]} for k, v in phases.items()}
        if one_line and "roofline" in one_line:
            # The roofline of the DOMINANT KERNEL is a statement about the kernel: it is taken from the launches that cover the whole
            # workload with the GPU to themselves -- the one-engine leg of this same command (same workload, co_shards = 1, its own
            # timed region of --steps ticks, HIP events riding on the launches; tools/profile_all.sh runs rocprofv3 on exactly that:
            # `bench.py --co-shards 1`).  The headline's own launches (one shard each, next to the other shard's) are listed under it.
            full = dict(one_line["roofline"])
            full["measured_in"] = ("one_engine leg of this command: co_shards = 1, %d timed ticks, %d launches sampled by events; "
                                   "the headline tick (`value`) is the co-scheduled one" % (one_line["steps"], one_line["launches_timed"]))
            full["co_scheduled_launch"] = {k: roofline.get(k) for k in ("kernel", "samples_per_launch", "avg_launch_us", "launches_timed", "achieved", "frac",
                                                                         "clock_mhz_under_load", "concurrent_launches", "concurrency_note", "traffic")}
            full["co_scheduled_launch"]["accounting_8d_frac"] = roofline["accounting_8d"]["frac"]
            full["headline_tick"] = {"tick_us": tick_s * 1e6, "tick_floor_us": full.get("tick_floor_us"),
                                     "tick_frac": (full["tick_floor_us"] / (tick_s * 1e6)) if full.get("tick_floor_us") else None,
                                     "note": "the one-engine floor against the co-scheduled headline tick"}
            roofline = full
        roofline["tick_level"] = tick_level
        roofline.update(extra_roof)
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # reported at N = 1 only
            cpu = cpu_baseline(T, goal if goal is not None else [0.0, -1.0, 0.0])
        value = units_total / (elapsed / args.steps)
        if args.storage == "f64":
            dtype, detail = "f64", "fp64 state, cost, storage and softmax (the reference's precision end to end)"
        elif roofline["kernel"] == "rollout_pk_kernel":
            dtype = "f32"
            detail = ("mixed: rollout deviations from the fp64 nominal trajectory in packed fp32 (clip, heading series, speed / Simpson-weight "
                      "deviations, noise cost) with the three running sums (heading, position, cost prefix), the rotation into the world frame "
                      "and the quadratic cost in fp64; noise drawn in fp32 (Philox + Box-Muller); HBM-resident cost prefix fp32 offsets from the "
                      "nominal trajectory; softmax weights (exp, sums) fp32; merge / control update / filter / plant step fp64.  "
                      "|V - V_oracle| <= 3e-7 max(1, |V - V_nominal|) + lambda / 100 (tests); the all-fp64 line is `f64_storage`")
        else:
            dtype = "f64"
            detail = ("rollout state + cost arithmetic fp64; noise drawn in fp32 (Philox + Box-Muller); HBM-resident cost prefix "
                      "fp32 offsets from the nominal trajectory; softmax weights (exp, sums) fp32; merge / control update / "
                      "filter / plant step fp64" if lanes else
                      "scan kernel: all arithmetic fp64 (noise drawn in fp32), nothing stored")
        line = {
            "metric": "MPPI rollouts/sec per control tick (K x T state steps)",
            "value": value, "unit": "rollouts/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None, "dtype": dtype, "dtype_detail": detail, "data": "synthetic",
            "config": {"workload": desc, "agents": A_total, "samples_total": K_total, "horizon": T,
                       "samples_per_gpu": K_local if args.workload != "c5" else K_total,
                       "state_steps_per_tick": units_total * T, "storage": args.storage,
                       "noise": "device Philox4x32-10", "sigma": 0.9, "lambda": 0.001,
                       "parallelism": ("K-sharded x%d, exchange: %s" % (world, exch_kind)) if args.workload != "c5" else "agent replicas",
                       "graph": bool(args.graph), "tick_kernels": info.get("tick_kernels", "lanes"),
                       "co_shards": info.get("co_shards", 1), "co_samples": info.get("co_samples"),
                       # what every rank's size rules were given (mppi_config.samples_total): the whole controller's samples, so that
                       # N = 1 / 2 / 4 / 8 run the same kernels and end every tick with the same controls to rounding; 0: each rank's own share
                       "kernels_pinned_by_samples_total": extra.get("samples_total_pinned", 0),
                       "min_warmup_s": args.min_warmup_s},
            "state_steps_per_s": value * T,
            "tick_us": tick_us,
            "final_state": [float(x) for x in final_nxt[0]], "final_u": [float(x) for x in final_ua[0]],
            "sync_tick_us": sync_tick_us,
            "kernels_us": kernels_us, "exchange_us": exchange_us, "per_rank": per_rank,
            "roofline": roofline, "cpu_baseline": cpu, "f64_storage": f64_line, "one_engine": one_line, "noise_packing_1": pack_line,
        }
        line.update(extra)
        # the other regime of the closed loop at the top level too: the robot parked at its goal
        if "parked_at_goal" in extra:
            line["value_parked_at_goal"] = extra["parked_at_goal"]["value"]
        # The full record goes to a side file; the line that is PRINTED is the contract's keys plus FLAT `roofline` / `cpu_baseline`
        # objects (scalars only: a record that keeps top-level scalars loses nothing) and stays under 4 KB.
        full_path = args.full_out
        if not full_path:
            base = os.environ.get("GRAFT_REPO_ROOT") or ROOT
            out_dir = os.path.join(base, "gpurun_out") if os.environ.get("GRAFT_REPO_ROOT") else os.path.join(base, "profiles")
            tag = args.workload + ("_K%d" % args.samples if args.samples else "") + ("_T%d" % args.horizon if args.horizon else "") + \
                ("" if args.storage == "f32" else "_" + args.storage) + ("" if args.co_shards is None else "_co%d" % args.co_shards) + \
                ("" if args.gpus == 1 else "_n%d" % args.gpus)
            full_path = os.path.join(out_dir, "bench_full_%s.json" % tag)
        try:
            os.makedirs(os.path.dirname(full_path), exist_ok=True)
            json.dump(line, open(full_path, "w"))
        except OSError:
            full_path = None
        out = compact_line(line, full_path)
        text = json.dumps(out)
        for drop in ("dtype_detail", "final_state", "final_u", "kernels_us", "per_rank"):   # (never needed so far: 3.1 KB at N = 1, 3.6 KB at N = 8)
            if len(text) <= 4096:
                break
            out.pop(drop, None)
            text = json.dumps(out)
        print(text)
    if in_group:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
