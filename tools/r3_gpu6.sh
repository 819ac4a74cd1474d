#!/bin/bash
mkdir -p gpurun_out/g6
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/g6/pytest_all.log 2>&1
tail -12 gpurun_out/g6/pytest_all.log
python - <<'PY' > gpurun_out/g6/one_engine.jsonl 2> gpurun_out/g6/one_engine.err
import sys, json, time, numpy as np
sys.path.insert(0, '.')
from motion_planning_amd.mppi import Engine
for K in (1000000, 125000):
  for parked in (False, True):
    with Engine(K, 50, co_shards=1) as e:
        T = 50
        u0 = np.zeros((2, T)) if parked else np.array([np.linspace(-2, 1, T), np.linspace(1.5, -1, T)])
        goal = np.array([[0.0, -1.0, 0.0]]); start = goal if parked else np.zeros((1, 3))
        e.set_nominal(u0); e.tick_async(start, goal, seed=0, tick_id=0)
        for i in range(1, 300): e.tick_async(seed=0, tick_id=i)
        e.set_nominal(u0); e.tick_async(start, goal, seed=0, tick_id=1000000); e.synchronize()
        t0 = time.perf_counter()
        for j in range(200): e.tick_async(seed=0, tick_id=1000001 + j)
        e.synchronize(); el = time.perf_counter() - t0
        e.kernel_timing(("rollout", "update", "merge", "finalize"), period=1)
        for j in range(40): e.tick_async(seed=0, tick_id=2000001 + j)
        e.synchronize()
        print(json.dumps({"K": K, "parked": parked, "tick_us": 1e6 * el / 200, "k": {k: round(1e3 * v[0] / max(v[1], 1), 2) for k, v in e.kernel_times().items() if v[1]}}), flush=True)
PY
cat gpurun_out/g6/one_engine.jsonl
timeout 600 python bench.py --steps 100 --no-cpu-baseline > gpurun_out/g6/bench_c4.json 2> gpurun_out/g6/bench_c4.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/g6/bench_c4.json').read().strip().splitlines()[-1])
print("value %.4g ms %.4f" % (d['value'], d['ms_per_step'])); print("parked", d.get('parked_at_goal')); o=d['one_engine']; print("one", o['ms_per_step'], o['kernels_us_bracketed'], o['parked_at_goal'], o['self_check']); print(d['sync_tick_us'])
PY
