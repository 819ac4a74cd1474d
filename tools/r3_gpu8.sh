#!/bin/bash
mkdir -p gpurun_out/g7
export TMPDIR=/tmp
run() { # label, env..., co
  python - "$@" <<'PY'
import sys, json, time, os, numpy as np
sys.path.insert(0, '.')
from motion_planning_amd.mppi import Engine
label, co = sys.argv[1], int(sys.argv[2])
K, T = 1000000, 50
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
    print(json.dumps({"label": label, "co": e.info()["co_samples"], "tick_us": [round(x, 2) for x in res]}), flush=True)
PY
}
{
MPPI_CO_CUT_PCT=58 run "co2 58/42" 2
MPPI_CO_CUT_PCT=60 run "co2 60/40" 2
MPPI_CO_CUT_PCT=62 run "co2 62/38" 2
MPPI_CO_CUT_PCT=50,80 run "co3 50/30/20" 3
MPPI_CO_CUT_PCT=45,80 run "co3 45/35/20" 3
MPPI_CO_CUT_PCT=55,85 run "co3 55/30/15" 3
MPPI_CO_CUT_PCT=45,75,92 run "co4 45/30/17/8" 4
} > gpurun_out/g7/co_cut2.jsonl 2> gpurun_out/g7/co_cut2.err
cat gpurun_out/g7/co_cut2.jsonl; tail -3 gpurun_out/g7/co_cut2.err
