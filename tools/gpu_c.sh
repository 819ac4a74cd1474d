#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c; mkdir -p $O
cd $R
( time timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -s 2>&1 ) > $O/test.log 2>&1
tail -25 $O/test.log | cut -c1-300
grep -E "full-size replay|f32 vs f64|config 5:|pentagon:|many weighted" $O/test.log | cut -c1-250
timeout 200 python tools/node_latency.py 2>&1 | tee $O/node_latency.txt
for w in c2 c4; do timeout 300 python bench.py --workload $w --no-cpu-baseline --no-f64-line > $O/bench_$w.json 2> $O/bench_$w.err; tail -c 300 $O/bench_$w.err; done
python - <<PY
import json
for f in ("bench_c2","bench_c4"):
    try:
        d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1])
        print(f, "ms/step %.4f"%d["ms_per_step"], "tick_us", d["tick_us"], "kernels", d["kernels_us"], "sync", d.get("sync_tick_us"), "parked", d.get("parked_at_goal"))
    except Exception as e: print(f, "ERR", e)
PY
