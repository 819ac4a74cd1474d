import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_amd import _capi
_capi.LIB_PATH = os.path.abspath(sys.argv[1])
from motion_planning_amd.mppi import Engine
K, T = 1000000, 50
co = None if len(sys.argv) > 2 else 1
with Engine(K, T, storage="f32", co_shards=co) as e:
    u0 = np.array([np.linspace(-2, 1, T), np.linspace(1.5, -1, T)])
    e.set_nominal(u0)
    e.tick_async([0, 0, 0], [0, -1, 0], seed=0, tick_id=0)
    t0, i = time.perf_counter(), 1
    while time.perf_counter() - t0 < 0.3:
        for _ in range(16):
            e.tick_async(seed=0, tick_id=i); i += 1
        e.synchronize()
    out = []
    for rep in range(3):
        e.set_nominal(u0); e.tick_async([0, 0, 0], [0, -1, 0], seed=0, tick_id=1000000); e.synchronize()
        t0 = time.perf_counter()
        for j in range(400):
            e.tick_async(seed=0, tick_id=1000001 + j)
        e.synchronize()
        out.append(round(1e6 * (time.perf_counter() - t0) / 400, 2))
    nxt, ua = e.get_outputs()
    e.kernel_timing(("rollout",), period=2)
    for j in range(200):
        e.tick_async(seed=0, tick_id=3000001 + j)
    e.synchronize()
    dt = e.kernel_times()
print(json.dumps({"lib": os.path.basename(sys.argv[1]), "co": co, "tick_us": out, "rollout_us": round(1e3 * dt["rollout"][0] / dt["rollout"][1], 2), "u": [float(x) for x in ua[0]]}), flush=True)
