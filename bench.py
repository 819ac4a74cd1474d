#!/usr/bin/env python3
"""bench.py -- MPPI rollouts/s per control tick on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c4|c2|c3|c5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full control tick (MPPI.get_path, control/src/mppi:85-102) on synthetic
inputs, device Philox noise, closed loop on the model (the predicted state feeds the next
tick), everything resident in HBM: nominal baseline -> rollout+cost -> per-timestep softmax
partials -> [RCCL all-gather of the [A][T][8] partials when K is sharded] -> control update,
clip, Savitzky-Golay, clip, plant step, shift.

Default workload = BASELINE config 4, the one the north-star target is quoted on:
parallel park, K = 1 000 000 rollouts, T = 50, K split over the N GPUs (strong scaling: the
north star "2/4/8-GPU runs split K").  It fits one GPU, so N = 1 runs all of it.
--workload c2 / c3 / c5 select the other BASELINE configs (single-GPU cases; c5 = 64 agents).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_PER_STEP_PER_KERNEL = 12  # eps 2 x fp32 + V 1 x fp32, written once (rollout) / read once (update)

WORKLOADS = {
    # name: (description, agents, K_total, T, goal)
    "c4": ("parallel-park K=1000000 T=50 (BASELINE config 4; K split over the GPUs)", 1, 1000000, 50, [0.0, -1.0, 0.0]),
    "c2": ("parallel-park K=10000 T=50 (BASELINE config 2)", 1, 10000, 50, [0.0, -1.0, 0.0]),
    "c3": ("pentagon waypoint-follow K=100000 T=100 (BASELINE config 3, first waypoint)", 1, 100000, 100, [1.0, 0.0, 0.0]),
    "c5": ("64 agents x K=16384 T=50 (BASELINE config 5; agents split over the GPUs)", 64, 16384, 50, None),
}


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


def cpu_baseline(T, goal, budget_s=12.0):
    """The oracle (C restatement of the reference loop, oracle/mppi_oracle.c) timed on this box's
    host cores on a bounded sample of the same workload: K_cpu rollouts of the same T, once on
    1 thread (the scalar port) and once with OpenMP over K on every usable core; the better
    of the two is `value` (with its thread count in `cores`)."""
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
            if el > budget_s / 2 or n >= 20:
                break
        out[used] = K_cpu * n / el
    best = max(out, key=lambda k: out[k])
    return {"value": out[best], "unit": "rollouts/s", "cores": best, "kind": "port",
            "by_threads": {str(k): v for k, v in out.items()}, "usable_cores": cores,
            "sample": "oracle get_path ticks (fp64, injected noise, OpenMP over K), K=%d T=%d, same goal/nominal as the "
                      "GPU workload; noise generation excluded" % (K_cpu, T)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--storage", default="f32", choices=["f32", "f64"])
    ap.add_argument("--samples", type=int, default=0, help="override K_total")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the tick as one hipGraph (N=1 only)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from motion_planning_amd import sharded

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with torch.distributed.run (one process per GPU)")
        args.gpus = world
    torch.cuda.set_device(local_rank)
    in_group = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ  # launched by torch.distributed.run
    if in_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    desc, A_total, K_total, T, goal = WORKLOADS[args.workload]
    if args.samples:
        K_total = args.samples
    if args.workload == "c5":  # independent agents: replicas, no collective
        lo, hi = sharded.shard_range(A_total, world, rank)
        A = hi - lo
        from motion_planning_amd.mppi import Engine
        eng = Engine(K_total, T, n_agents=A, storage=args.storage, device=local_rank)
        ticker = sharded.ShardedTicker.__new__(sharded.ShardedTicker)
        ticker.shard = sharded.HipShard(eng, torch.device("cuda", local_rank), use_torch_stream=False)
        ticker.world, ticker.rank, ticker.dist, ticker.group, ticker._gathered = 1, 0, None, None, None
        states = np.array([[0.05 * a, 0.0, 0.0] for a in range(lo, hi)])
        goals = np.array([[0.05 * a, -1.0, 0.0] for a in range(lo, hi)])
        K_local, units_total = K_total, A_total * K_total
    else:
        ticker, eng = sharded.make_hip_ticker(K_total, T, n_agents=1, storage=args.storage, local_rank=local_rank)
        A = 1
        states, goals = np.zeros((1, 3)), np.array([goal])
        K_local, units_total = eng.K, K_total
    for a in range(A):
        eng.set_nominal(nominal_warm(T), agent=a)

    seed = 0
    def tick(i, first=False):
        if args.graph and not in_group and not first:
            eng.tick_graph(seed)
        else:
            ticker.tick_async(states if first else None, goals if first else None, "philox", seed, i)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    tick(0, first=True)
    for i in range(1, args.warmup + 1):
        tick(i)
    # Timed region: exactly --steps ticks between barrier + synchronize pairs.  The dominant kernel
    # (rollout) is timed live with HIP events that ride on its own launch (hipExtLaunchKernelGGL start /
    # stop events on the engine's stream = the dispatch's begin / end timestamps, the clock rocprofv3
    # --kernel-trace reads); no marker packets enter the stream, so every EVENT_PERIOD-th launch is sampled
    # only to keep the event pool small.
    EVENT_PERIOD = int(os.environ.get("MPPI_EVENT_PERIOD", "4"))
    eng.kernel_timing(("rollout",), period=EVENT_PERIOD)
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        tick(args.warmup + 1 + i)
    sync()
    elapsed = time.perf_counter() - t0
    ktimes = eng.kernel_times()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps

    # Diagnostic pass (not the headline clock): every kernel bracketed on every launch.
    n_diag = min(args.steps, 20)
    eng.kernel_timing(("nominal", "rollout", "update", "merge", "finalize"), period=1)
    sync()
    for i in range(n_diag):
        tick(args.warmup + 1 + args.steps + i)
    sync()
    dtimes = eng.kernel_times()
    eng.kernel_timing(())
    nxt, ua = eng.get_outputs()
    assert np.isfinite(nxt).all() and np.isfinite(ua).all()
    final_nxt, final_ua = nxt.copy(), ua.copy()

    # Diagnostic: the node's own call pattern -- host state in, blocking, host controls out
    # (mppi_tick; what Controller.pos_cb pays per odometry message), N = 1 only.
    sync_tick_us = None
    btimes = None
    if not in_group:
        n_lat = min(args.steps, 200)
        eng.kernel_timing(("rollout",), period=1)
        st, lat = nxt, []
        for i in range(n_lat):
            t0 = time.perf_counter()
            st, _ = eng.tick(st, goals, noise="philox", seed=seed, tick_id=10_000_000 + i)
            lat.append(1e6 * (time.perf_counter() - t0))
        btimes = eng.kernel_times()
        eng.kernel_timing(())
        lat = np.sort(np.array(lat))
        sync_tick_us = {"mean": float(lat.mean()), "median": float(np.median(lat)),
                        "p99": float(lat[min(len(lat) - 1, int(0.99 * len(lat)))]), "ticks": n_lat}

    if rank == 0:
        steps_per_launch = A * K_local * T
        ms, n = ktimes["rollout"]
        if n == 0:  # hipGraph replay: launches are not individually bracketed
            ms, n = dtimes["rollout"] if dtimes["rollout"][1] else (float("nan"), 1)
        avg_s = ms * 1e-3 / max(n, 1)
        gbs = BYTES_PER_STEP_PER_KERNEL * steps_per_launch / avg_s / 1e9
        tick_s = elapsed / args.steps
        roofline = {"kernel": "rollout_kernel", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                    "algorithmic_bytes_per_launch": BYTES_PER_STEP_PER_KERNEL * steps_per_launch,
                    "avg_launch_us": avg_s * 1e6, "launches_timed": n, "event_period": EVENT_PERIOD,
                    "note": "algorithmic = 12 B/state-step per kernel (eps 2xfp32 + V fp32, SURVEY 8d: 24 B/step per tick, "
                            "written once by rollout, read once by update); the kernel itself is VALU-bound (fp64 state "
                            "+ Philox) and, in the tick path, writes only 4 of those 12 B (eps is regenerated, not stored)",
                    "tick_level": {"algorithmic_bytes": 2 * BYTES_PER_STEP_PER_KERNEL * steps_per_launch,
                                   "achieved": 2 * BYTES_PER_STEP_PER_KERNEL * steps_per_launch / tick_s / 1e9,
                                   "frac": 2 * BYTES_PER_STEP_PER_KERNEL * steps_per_launch / tick_s / 1e9 / HBM_PEAK_GBS}}
        # HBM bytes actually moved per launch of that kernel: rocprofv3 --pmc passes of this same command
        # (tools/pmc.sh), summary committed under profiles/ -- bench.py itself cannot host the profiler
        pmc_file = os.path.join(ROOT, "profiles", "r1_pmc_summary_bench_c4.json")
        if args.workload == "c4" and args.storage == "f32" and world == 1 and not args.samples and os.path.exists(pmc_file):
            pm = json.load(open(pmc_file))
            for kname, c in pm.items():
                if "rollout_kernel" in kname:
                    rd = [v for k, v in c.items() if k.startswith("hbm_read_bytes")]
                    wr = [v for k, v in c.items() if k.startswith("hbm_write_bytes")]
                    if rd and wr:
                        roofline["traffic"] = rd[0] + wr[0]
                        roofline["traffic_source"] = "profiles/r1_pmc_summary_bench_c4.json (FETCH_SIZE x2 + WRITE_SIZE, bytes/launch)"
        # the rollout launch in each phase of this command (what a rocprofv3 --kernel-trace --stats of the whole
        # command averages over): back-to-back ticks run ~8 % longer than launches behind an idle gap
        phases = {"timed": ktimes["rollout"], "diagnostic": dtimes["rollout"]}
        if btimes:
            phases["blocking"] = btimes["rollout"]
        roofline["rollout_us_by_phase"] = {k: {"avg_us": (v[0] * 1e3 / v[1] if v[1] else None), "launches_timed": v[1]}
                                           for k, v in phases.items()}
        kernels_us = {name: (dtimes[name][0] * 1e3 / dtimes[name][1] if dtimes[name][1] else None) for name in dtimes}
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # reported at N = 1 only
            cpu = cpu_baseline(T, goal if goal is not None else [0.0, -1.0, 0.0])
        value = units_total / (elapsed / args.steps)
        line = {
            "metric": "MPPI rollouts/sec per control tick (K x T state steps)",
            "value": value, "unit": "rollouts/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc, "agents": A_total, "samples_total": K_total, "horizon": T,
                       "samples_per_gpu": K_local if args.workload != "c5" else K_total,
                       "state_steps_per_tick": units_total * T, "storage": args.storage,
                       "noise": "device Philox4x32-10", "sigma": 0.9, "lambda": 0.001,
                       "parallelism": ("K-sharded x%d + all-gather" % world) if args.workload != "c5" else "agent replicas",
                       "graph": bool(args.graph), "tick_kernels": eng.info()["tick_kernels"]},
            "state_steps_per_s": value * T,
            "final_state": [float(x) for x in final_nxt[0]], "final_u": [float(x) for x in final_ua[0]],
            "sync_tick_us": sync_tick_us,
            "kernels_us": kernels_us, "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if in_group:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
