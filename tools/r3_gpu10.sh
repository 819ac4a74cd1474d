#!/bin/bash
mkdir -p gpurun_out/g10
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_p2p_gpu.py -m gpu -q -x -k "not large_k" > gpurun_out/g10/pytest_p2p.log 2>&1
tail -5 gpurun_out/g10/pytest_p2p.log
run() {
  python - "$@" <<'PY'
import sys, json, time, os, numpy as np
sys.path.insert(0, '.')
from motion_planning_amd.mppi import Engine
label, co, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
T = 50
with Engine(K, T, co_shards=co) as e:
    u0 = np.array([np.linspace(-2, 1, T), np.linspace(1.5, -1, T)]); goal = np.array([[0.0, -1.0, 0.0]])
    e.set_nominal(u0); e.tick_async(np.zeros((1, 3)), goal, seed=0, tick_id=0)
    t0 = time.perf_counter(); i = 1
    while time.perf_counter() - t0 < 0.4:
        for _ in range(16): e.tick_async(seed=0, tick_id=i); i += 1
        e.synchronize()
    res = []
    for rep in range(3):
        e.set_nominal(u0); e.tick_async(np.zeros((1, 3)), goal, seed=0, tick_id=1000000); e.synchronize()
        t0 = time.perf_counter()
        for j in range(200): e.tick_async(seed=0, tick_id=1000001 + j)
        e.synchronize(); res.append(1e6 * (time.perf_counter() - t0) / 200)
    nxt, ua = e.get_outputs()
    print(json.dumps({"label": label, "co": e.info()["co_samples"], "tick_us": [round(x, 2) for x in res], "u": [float(x) for x in ua[0]]}), flush=True)
PY
}
{
run "co2 fused" 0 1000000
MPPI_DEFER_MERGE=0 run "co2 separate" 0 1000000
run "co2 fused" 0 1000000
MPPI_DEFER_MERGE=0 run "co2 separate" 0 1000000
run "co1" 1 1000000
run "co2 fused 600k" 0 600000
MPPI_DEFER_MERGE=0 run "co2 separate 600k" 0 600000
} > gpurun_out/g10/merge_ab.jsonl 2> gpurun_out/g10/merge_ab.err
cat gpurun_out/g10/merge_ab.jsonl; tail -3 gpurun_out/g10/merge_ab.err
