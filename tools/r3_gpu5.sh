#!/bin/bash
mkdir -p gpurun_out/g5
export TMPDIR=/tmp
g++ -O2 -std=c++17 -Iinclude tools/node_tail.cpp -o tools/node_tail -Lmotion_planning_amd/lib -lmppi_hip -Wl,-rpath,$PWD/motion_planning_amd/lib 2> gpurun_out/g5/node_tail_build.err
( ./tools/node_tail 10 100 5000 0; ./tools/node_tail 10 100 3000 500; ./tools/node_tail 1000 50 3000 0; ./tools/node_tail 10000 50 3000 0 ) > gpurun_out/g5/node_tail.txt 2>&1
head -c 3000 gpurun_out/g5/node_tail.txt
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/g5/pytest_all.log 2>&1
tail -12 gpurun_out/g5/pytest_all.log
python tools/ab_rollout.py --rounds 1 --parked > gpurun_out/g5/ab.jsonl 2> gpurun_out/g5/ab.err
MPPI_CO=1 python - <<'PY' >> gpurun_out/g5/ab.jsonl 2>> gpurun_out/g5/ab.err
import sys, json, time, numpy as np
sys.path.insert(0, '.')
from motion_planning_amd.mppi import Engine
# one-engine parked / underway update kernel with the queue (co_shards = 1)
for parked in (False, True):
    with Engine(1000000, 50, co_shards=1) as e:
        T = 50
        u0 = np.zeros((2, T)) if parked else np.array([np.linspace(-2, 1, T), np.linspace(1.5, -1, T)])
        goal = np.array([[0.0, -1.0, 0.0]]); start = goal if parked else np.zeros((1, 3))
        e.set_nominal(u0); e.tick_async(start, goal, seed=0, tick_id=0)
        for i in range(1, 300): e.tick_async(seed=0, tick_id=i)
        e.set_nominal(u0); e.tick_async(start, goal, seed=0, tick_id=1000000)
        e.kernel_timing(("rollout", "update", "merge", "finalize"), period=1); e.synchronize()
        t0 = time.perf_counter()
        for j in range(100): e.tick_async(seed=0, tick_id=1000001 + j)
        e.synchronize(); el = time.perf_counter() - t0
        print(json.dumps({"one_engine": True, "parked": parked, "tick_us": 1e4 * el, "k": {k: 1e3 * v[0] / max(v[1], 1) for k, v in e.kernel_times().items() if v[1]}}))
PY
cat gpurun_out/g5/ab.jsonl | cut -c1-400
