"""Run ON THE GPU BOX: the fp64-storage tick as ONE fused kernel (rollout_fused.hpp) against rollout + update, the headline's protocol
(0.3 s of warm-up ticks, controller back at the start, 100 timed ticks), engines alternating, two rounds, four sizes (EXPERIMENTS.md 58)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_amd.mppi import Engine
T = 50
for K in (1000000, 500000, 250000, 125000):
    for rnd in range(2):
        for name, opts in (("fused", {"pk_min_samples": 1}), ("two-kernel", {"rollout_pk": 0})):
            with Engine(K, T, storage="f64", tick_path="lanes", options=opts) as e:
                e.set_nominal(np.array([np.linspace(-2, 1, T), np.linspace(1.5, -1, T)]))
                e.tick([0, 0, 0], [0, -1, 0], noise="philox", seed=0, tick_id=0)
                t0 = time.perf_counter(); i = 1
                while time.perf_counter() - t0 < 0.3:
                    e.tick_async(None, None, noise="philox", seed=0, tick_id=i); i += 1
                    if i % 16 == 0: e.synchronize()
                e.set_nominal(np.array([np.linspace(-2, 1, T), np.linspace(1.5, -1, T)]))
                e.tick([0, 0, 0], [0, -1, 0], noise="philox", seed=0, tick_id=1000000)
                e.synchronize()
                t0 = time.perf_counter()
                for i in range(100): e.tick_async(None, None, noise="philox", seed=0, tick_id=1000001 + i)
                e.synchronize()
                el = time.perf_counter() - t0
                print("K %7d %-10s tick us %.1f  (%s)" % (K, name, 1e4 * el, e.info()["rollout_kernel"]), flush=True)
