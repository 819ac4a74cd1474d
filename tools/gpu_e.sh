#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/e; mkdir -p $O
cd $R
timeout 300 python tools/p2p_debug.py 2>&1 | tail -12 | cut -c1-300
( time timeout 1200 python -m pytest tests -m gpu -q --timeout 300 2>&1 ) > $O/test.log 2>&1
tail -15 $O/test.log | cut -c1-300
for k in 500000 250000 125000; do timeout 200 python bench.py --samples $k --no-cpu-baseline --no-f64-line --steps 100 > $O/bench_k$k.json 2>/dev/null; done
python - <<PY
import json
for k in (500000,250000,125000):
    try:
        d=json.loads(open("$O/bench_k%d.json"%k).read().strip().splitlines()[-1])
        print(k, "ms/step %.4f"%d["ms_per_step"], "tick_us", d["tick_us"]["median"], "kernels", d["kernels_us"])
    except Exception as e: print(k, "ERR", e)
PY
