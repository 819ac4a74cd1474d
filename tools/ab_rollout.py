#!/usr/bin/env python3
"""Same-box A/B of the two tick-path rollout kernels of an fp32-storage engine: option "rollout_pk" = 0 (rollout_kernel, all
fp64) against = 1 (rollout_pk_kernel, mixed precision, two samples per lane), alternating, config 4 by default.

    python tools/ab_rollout.py [--samples K] [--horizon T] [--rounds 3] [--ticks 300]
Prints one JSON line per (round, kernel): tick time, the rollout launch's mean duration (events riding on the launch),
update kernel duration, and the shader clock a probe wave inside the rollout launch measured."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_amd.mppi import Engine


def run(pk, K, T, ticks, parked=False, co=1):
    with Engine(K, T, storage="f32", tick_path="lanes", co_shards=co, options={"rollout_pk": int(pk)}) as e:
        u0 = np.zeros((2, T)) if parked else np.array([np.linspace(-2, 1, T), np.linspace(1.5, -1, T)])
        goal = np.array([[0.0, -1.0, 0.0]])
        start = goal if parked else np.zeros((1, 3))
        e.set_nominal(u0)
        e.tick_async(start, goal, seed=0, tick_id=0)
        t0, i = time.perf_counter(), 1
        while time.perf_counter() - t0 < 0.4:
            for _ in range(16):
                e.tick_async(seed=0, tick_id=i); i += 1
            e.synchronize()
        e.set_nominal(u0)
        e.tick_async(start, goal, seed=0, tick_id=1000000)
        e.kernel_timing(("rollout",), period=4)
        e.synchronize()
        t0 = time.perf_counter()
        for j in range(ticks):
            e.tick_async(seed=0, tick_id=1000001 + j)
        e.synchronize()
        el = time.perf_counter() - t0
        kt = e.kernel_times()
        mhz = e.shader_clock_mhz()
        nxt, ua = e.get_outputs()
        e.kernel_timing(("rollout", "update", "merge", "finalize"), period=1)
        for j in range(20):
            e.tick_async(seed=0, tick_id=2000001 + j)
        e.synchronize()
        dt = e.kernel_times()
    return {"pk": int(pk), "K": K, "T": T, "parked": parked, "tick_us": 1e6 * el / ticks,
            "rollout_us": 1e3 * kt["rollout"][0] / max(kt["rollout"][1], 1), "shader_mhz": mhz,
            "bracketed_us": {k: 1e3 * v[0] / max(v[1], 1) for k, v in dt.items() if v[1]},
            "u_applied": [float(x) for x in ua[0]], "state": [float(x) for x in nxt[0]]}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=1000000)
    ap.add_argument("--horizon", type=int, default=50)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--ticks", type=int, default=300)
    ap.add_argument("--parked", action="store_true")
    ap.add_argument("--co-shards", type=int, default=1, help="1: one engine (the kernels undisturbed); 0: the engine's own rule")
    a = ap.parse_args()
    for r in range(a.rounds):
        for pk in (0, 1):
            print(json.dumps(dict(run(pk, a.samples, a.horizon, a.ticks, a.parked, a.co_shards), round=r, co_shards=a.co_shards)), flush=True)
