#!/usr/bin/env python3
"""A plain loop of config-4 ticks in one of the three tick forms (option "fused": 0 stand-alone kernels, 1 one fused launch, 2 update
stream), for rocprofv3 passes (tools/pmc_fused.sh).  Prints the tick time."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_amd.mppi import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--fused", type=int, default=0)
ap.add_argument("--samples", type=int, default=1000000)
ap.add_argument("--ticks", type=int, default=40)
ap.add_argument("--lag", type=int, default=0)
ap.add_argument("--tail-blocks", type=int, default=2048)
a = ap.parse_args()
T = 50
with Engine(a.samples, T, storage="f32", tick_path="lanes", co_shards=1,
            options={"fused": a.fused, "fused_lag": a.lag, "us_tail_blocks": a.tail_blocks}) as e:
    e.set_nominal(np.array([np.linspace(-2, 1, T), np.linspace(1.5, -1, T)]))
    e.tick_async(np.zeros((1, 3)), np.array([[0.0, -1.0, 0.0]]), "philox", 0, 0)
    for i in range(1, 10):
        e.tick_async(None, None, "philox", 0, i)
    e.synchronize()
    t0 = time.perf_counter()
    for i in range(a.ticks):
        e.tick_async(None, None, "philox", 0, 100 + i)
    e.synchronize()
    print(json.dumps({"fused": a.fused, "samples": a.samples, "tick_us": 1e6 * (time.perf_counter() - t0) / a.ticks, "ran": e.info()["tick_fused"]}))
