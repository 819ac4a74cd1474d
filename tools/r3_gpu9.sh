#!/bin/bash
mkdir -p gpurun_out/g9
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_p2p_gpu.py tests/test_gpu_parity.py -m gpu -q -x -k "co_scheduled or p2p_two or shard_partials or update_after or multi_agent or config2 or odd_sizes or long_horizons or tick_graph or full_size" > gpurun_out/g9/pytest_sub.log 2>&1
tail -6 gpurun_out/g9/pytest_sub.log
python - <<'PY' > gpurun_out/g9/sizes.jsonl 2> gpurun_out/g9/sizes.err
import sys, json, time, numpy as np
sys.path.insert(0, '.')
from motion_planning_amd.mppi import Engine
T = 50
for K, co in ((1000000, 0), (1000000, 1), (500000, 1), (250000, 1), (125000, 1), (500000, 0)):
    with Engine(K, T, co_shards=co) as e:
        u0 = np.array([np.linspace(-2, 1, T), np.linspace(1.5, -1, T)]); goal = np.array([[0.0, -1.0, 0.0]])
        e.set_nominal(u0); e.tick_async(np.zeros((1, 3)), goal, seed=0, tick_id=0)
        t0 = time.perf_counter(); i = 1
        while time.perf_counter() - t0 < 0.3:
            for _ in range(16): e.tick_async(seed=0, tick_id=i); i += 1
            e.synchronize()
        res = []
        for rep in range(3):
            e.set_nominal(u0); e.tick_async(np.zeros((1, 3)), goal, seed=0, tick_id=1000000); e.synchronize()
            t0 = time.perf_counter()
            for j in range(200): e.tick_async(seed=0, tick_id=1000001 + j)
            e.synchronize(); res.append(1e6 * (time.perf_counter() - t0) / 200)
        e.kernel_timing(("rollout", "update", "merge", "finalize", "exchange"), period=1)
        for j in range(30): e.tick_async(seed=0, tick_id=2000001 + j)
        e.synchronize()
        print(json.dumps({"K": K, "co": e.info()["co_samples"], "tick_us": [round(x, 2) for x in res], "k": {k: round(1e3 * v[0] / max(v[1], 1), 2) for k, v in e.kernel_times().items() if v[1]}}), flush=True)
PY
cat gpurun_out/g9/sizes.jsonl; tail -2 gpurun_out/g9/sizes.err
