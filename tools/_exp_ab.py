#!/usr/bin/env python3
"""(temporary) one library per process: python tools/_exp_ab.py <lib.so> [--parked]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_amd import _capi
_capi.LIB_PATH = os.path.abspath(sys.argv[1])
from motion_planning_amd.mppi import Engine
parked = "--parked" in sys.argv
K, T = 1000000, 50
with Engine(K, T, storage="f32", tick_path="lanes", co_shards=1) as e:
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
    out = []
    for rep in range(3):
        e.set_nominal(u0)
        e.tick_async(start, goal, seed=0, tick_id=1000000)
        e.synchronize()
        t0 = time.perf_counter()
        for j in range(300):
            e.tick_async(seed=0, tick_id=1000001 + j)
        e.synchronize()
        out.append(1e6 * (time.perf_counter() - t0) / 300)
    e.set_nominal(u0)
    e.tick_async(start, goal, seed=0, tick_id=1000000)
    e.kernel_timing(("rollout", "update", "merge", "finalize"), period=1)
    for j in range(40):
        e.tick_async(seed=0, tick_id=2000001 + j)
    e.synchronize()
    dt = e.kernel_times()
print(json.dumps({"lib": os.path.basename(sys.argv[1]), "parked": parked, "tick_us": out,
                  "bracketed_us": {k: round(1e3 * v[0] / max(v[1], 1), 2) for k, v in dt.items() if v[1]}}), flush=True)
