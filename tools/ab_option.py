#!/usr/bin/env python3
"""Same-box A/B of ONE engine option (include/mppi_hip.h, mppi_set_option): the same workload ticked with each value in turn,
alternating over the rounds so that box-to-box and minute-to-minute clock differences cancel.

    python tools/ab_option.py --option tail_fused --values 0,-1 [--samples K] [--horizon T] [--agents A] [--co-shards 1]
                              [--storage f32] [--tick-path auto] [--rounds 3] [--ticks 400]
One JSON line per (round, value): fused mppi_tick time (host clock over `ticks` back-to-back ticks), the bracketed kernel
durations, and the outputs (which must not differ between the values of an option that only changes the schedule)."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_amd import _capi
if os.environ.get("MPPI_AB_LIB"):   # a measurement build (make VARIANT=name): lib/libmppi_hip_<name>.so instead of the product library
    _capi.LIB_PATH = os.path.join(os.path.dirname(_capi.LIB_PATH), "libmppi_hip_%s.so" % os.environ["MPPI_AB_LIB"])
from motion_planning_amd.mppi import Engine


def run(opt, val, a):
    T, A = a.horizon, a.agents
    with Engine(a.samples, T, n_agents=A, storage=a.storage, tick_path=a.tick_path, co_shards=a.co_shards, samples_total=getattr(a, "samples_total", 0),
                options=dict(a.fixed_options, **{opt: int(val)})) as e:
        u0 = np.tile(np.array([np.linspace(-2, 1, T), np.linspace(1.5, -1, T)]), (A, 1, 1))
        goal = np.tile(np.array([[0.0, -1.0, 0.0]]), (A, 1))
        start = np.zeros((A, 3))
        e.set_nominal_all(u0)
        e.tick_async(start, goal, seed=0, tick_id=0)
        t0, i = time.perf_counter(), 1
        while time.perf_counter() - t0 < 0.4:
            for _ in range(16):
                e.tick_async(seed=0, tick_id=i); i += 1
            e.synchronize()
        e.set_nominal_all(u0)
        e.tick_async(start, goal, seed=0, tick_id=1000000)
        e.synchronize()
        t0 = time.perf_counter()
        for j in range(a.ticks):
            e.tick_async(seed=0, tick_id=1000001 + j)
        e.synchronize()
        el = time.perf_counter() - t0
        nxt, ua = e.get_outputs()
        e.kernel_timing(("rollout", "update", "merge", "finalize"), period=1)
        for j in range(20):
            e.tick_async(seed=0, tick_id=2000001 + j)
        e.synchronize()
        dt = e.kernel_times()
    return {"option": opt, "value": int(val), "fixed": a.fixed_options, "storage": a.storage, "K": a.samples, "T": T, "A": A, "tick_us": 1e6 * el / a.ticks,
            "bracketed_us": {k: 1e3 * v[0] / max(v[1], 1) for k, v in dt.items() if v[1]},
            "u_applied": [float(x) for x in np.asarray(ua).ravel()[:4]], "state": [float(x) for x in np.asarray(nxt).ravel()[:6]]}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--option", required=True)
    ap.add_argument("--values", default="0,1")
    ap.add_argument("--samples", type=int, default=1000000)
    ap.add_argument("--horizon", type=int, default=50)
    ap.add_argument("--agents", type=int, default=1)
    ap.add_argument("--storage", default="f32")
    ap.add_argument("--tick-path", default="auto")
    ap.add_argument("--co-shards", type=int, default=1)
    ap.add_argument("--samples-total", type=int, default=0, help="the engine is one share of a controller of this many samples (mppi_config.samples_total)")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--ticks", type=int, default=400)
    ap.add_argument("--fixed", default="", help="name=value,...: options every run gets next to the one under test")
    a = ap.parse_args()
    a.fixed_options = dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.fixed.split(",") if kv)
    for r in range(a.rounds):
        for v in a.values.split(","):
            print(json.dumps(dict(run(a.option, int(v), a), round=r, co_shards=a.co_shards)), flush=True)
