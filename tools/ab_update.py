#!/usr/bin/env python3
"""Same-box A/B of the update kernel's GROUPS (option "upd_group": 0 = one chunk per workgroup + the merge launch, round 3;
-1 = ceil(chunks / 16) chunks per workgroup, no merge launch) over the bench's sizes.

    python tools/ab_update.py [--rounds 2] [--ticks 200] [--configs c4,c4co,s500k,s250k,c5,c4f64]
One JSON line per (config, mode, round): tick time, bracketed kernel times, and the controls the run ended in (the two modes must
agree to the split-invariance bound of the tuple merge)."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_amd.mppi import Engine

CONFIGS = {"c4": (1000000, 50, 1, 1, "f32"), "c4co": (1000000, 50, 1, None, "f32"), "s500k": (500000, 50, 1, 1, "f32"), "s250k": (250000, 50, 1, 1, "f32"),
           "s125k": (125000, 50, 1, 1, "f32"), "c5": (16384, 50, 64, 1, "f32"), "c4f64": (1000000, 50, 1, 1, "f64"), "c3": (100000, 100, 1, 1, "f32")}


def run(K, T, A, co, storage, mode, ticks, parked=False):
    with Engine(K, T, n_agents=A, storage=storage, tick_path="lanes", co_shards=co, options={"upd_group": mode}) as e:
        u0 = np.zeros((2, T)) if parked else np.array([np.linspace(-2, 1, T), np.linspace(1.5, -1, T)])
        goal = np.tile([0.0, -1.0, 0.0], (A, 1))
        start = goal if parked else np.zeros((A, 3))
        for a in range(A):
            e.set_nominal(u0, agent=a)
        e.tick_async(start, goal, seed=0, tick_id=0)
        t0, i = time.perf_counter(), 1
        while time.perf_counter() - t0 < 0.3:
            for _ in range(16):
                e.tick_async(seed=0, tick_id=i); i += 1
            e.synchronize()
        for a in range(A):
            e.set_nominal(u0, agent=a)
        e.tick_async(start, goal, seed=0, tick_id=1000000)
        e.synchronize()
        t0 = time.perf_counter()
        for j in range(ticks):
            e.tick_async(seed=0, tick_id=1000001 + j)
        e.synchronize()
        el = time.perf_counter() - t0
        nxt, ua = e.get_outputs()
        e.kernel_timing(("rollout", "update", "merge", "finalize"), period=1)
        for j in range(20):
            e.tick_async(seed=0, tick_id=2000001 + j)
        e.synchronize()
        dt = e.kernel_times()
        info = e.info()
    return {"upd_group": mode, "K": K, "T": T, "A": A, "co_shards": info["co_shards"], "storage": storage, "parked": parked, "tick_us": 1e6 * el / ticks,
            "update_blocks": info["update_blocks"], "bracketed_us": {k: 1e3 * v[0] / max(v[1], 1) for k, v in dt.items() if v[1]},
            "u_applied": [float(x) for x in ua[0]], "state": [float(x) for x in nxt[0]]}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--ticks", type=int, default=200)
    ap.add_argument("--configs", default="c4,c4co,s500k,s250k,c5,c4f64")
    ap.add_argument("--modes", default="0,-1")
    ap.add_argument("--parked", action="store_true")
    a = ap.parse_args()
    for name in a.configs.split(","):
        K, T, A, co, storage = CONFIGS[name]
        for r in range(a.rounds):
            ref = None
            for mode in [int(m) for m in a.modes.split(",")]:
                out = run(K, T, A, co, storage, mode, a.ticks, a.parked)
                if ref is None:
                    ref = out
                out["max_abs_diff_u_vs_first_mode"] = float(np.abs(np.array(out["u_applied"]) - np.array(ref["u_applied"])).max())
                print(json.dumps(dict(out, config=name, round=r)), flush=True)
