"""Run ON THE GPU BOX: the co-scheduled config-4 tick from handle instance to handle instance.  Five handles are created one after the other in ONE
process (each: 0.4 s of warm-up ticks, then 3 x 300 back-to-back ticks on the host clock), the same with co_shards = 1 next to each.
MPPI_AB_LIB=name runs a measurement build (lib/libmppi_hip_<name>.so: make VARIANT=name EXTRA=...).  (EXPERIMENTS.md 54, 56)"""
import sys, time, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_amd import _capi
if os.environ.get('MPPI_AB_LIB'): _capi.LIB_PATH = os.path.join(os.path.dirname(_capi.LIB_PATH), 'libmppi_hip_%s.so' % os.environ['MPPI_AB_LIB'])
from motion_planning_amd.mppi import Engine
T = 50
def measure(co, opts=None):
    with Engine(1000000, T, co_shards=co, options=opts) as e:
        u0 = np.array([np.linspace(-2, 1, T), np.linspace(1.5, -1, T)])
        e.set_nominal(u0)
        e.tick_async(np.zeros((1, 3)), np.array([[0.0, -1.0, 0.0]]), seed=0, tick_id=0)
        t0 = time.perf_counter(); i = 1
        while time.perf_counter() - t0 < 0.4:
            for _ in range(16): e.tick_async(seed=0, tick_id=i); i += 1
            e.synchronize()
        e.set_nominal(u0); e.tick_async(np.zeros((1, 3)), np.array([[0.0, -1.0, 0.0]]), seed=0, tick_id=10**6); e.synchronize()
        out = []
        for rep in range(3):
            t0 = time.perf_counter()
            for j in range(300): e.tick_async(seed=0, tick_id=10**6 + 1 + rep * 300 + j)
            e.synchronize()
            out.append(round(1e6 * (time.perf_counter() - t0) / 300, 1))
        return out
for k in range(5):
    print(os.environ.get("MPPI_AB_LIB", "default"), "co", measure(0), "one", measure(1), flush=True)
