#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/b; mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests -m gpu -q --timeout 600 -s 2>&1 ) > $O/test.log 2>&1
tail -12 $O/test.log
grep -E "full-size replay|f32 vs f64|config 5:|pentagon:|many weighted" $O/test.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; tail -c 600 $O/bench_driver.err
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.err
python - <<PY
import json
for f in ("bench_driver","bench_default"):
    try:
        d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1])
        print(f, "ms/step %.4f"%d["ms_per_step"], "tick_us", d["tick_us"], "kernels", d["kernels_us"], "roof", d["roofline"]["frac"], d["roofline"].get("valu",{}).get("frac"), "f64", d.get("f64_storage"), "parked", d.get("parked_at_goal"), "sync", d.get("sync_tick_us"), "warm", d.get("warmup_ticks_run"))
    except Exception as e: print(f, "ERR", e)
PY
timeout 120 ./tools/ubench > $O/ubench.txt 2>&1; grep "wps=4" $O/ubench.txt
timeout 200 python tools/node_latency.py 2>&1 | tee $O/node_latency.txt
