#!/bin/bash
# fresh-process stress of the agent-split tick (a handle of 64 agents, lazily built second engine, pulls and pushes in between):
# N processes, each under a timeout; prints hangs / failures.   usage: tools/agents_stress.sh [N=40]
N=${1:-40}; R=$(cd "$(dirname "$0")/.." && pwd); cd $R
hang=0; fail=0
for i in $(seq 1 $N); do
  timeout 60 python - <<'PY' > /tmp/agents_stress_$i.log 2>&1
import numpy as np
from motion_planning_amd.mppi import Engine
A, K, T = 64, 16384, 50
rng = np.random.RandomState(1)
with Engine(K, T, n_agents=A) as e:
    assert e.info()["co_shards"] == 2
    st = rng.uniform(-0.2, 0.2, (A, 3)); goal = rng.uniform(-1, 1, (A, 3))
    for i in range(12):
        st, ua = e.tick(st, goal if i == 0 else None, noise="philox", seed=3, tick_id=i)
        if i % 4 == 3:
            e.get_nominal(A - 1); e.set_nominal(np.zeros((2, T)), agent=i % A)
    for i in range(12, 40):
        e.tick_async(None, None, noise="philox", seed=3, tick_id=i)
    st, ua = e.get_outputs()
    assert np.isfinite(st).all() and np.isfinite(ua).all()
print("ok")
PY
  rc=$?
  if [ $rc -eq 124 ]; then hang=$((hang+1)); elif [ $rc -ne 0 ]; then fail=$((fail+1)); tail -3 /tmp/agents_stress_$i.log; fi
done
echo "agent-split stress: $hang hangs, $fail failures in $N fresh processes"
