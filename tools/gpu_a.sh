#!/bin/bash
# round-2 GPU call A: full GPU test-suite, driver-like bench, default bench, hang hunt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/a; mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests -m gpu -q --timeout 600 -s 2>&1 ) > $O/test.log 2>&1
tail -30 $O/test.log
grep -E "full-size replay|f32 vs f64|config 5:|pentagon:" $O/test.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; tail -c 600 $O/bench_driver.err
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.err
python - <<PY
import json
for f in ("bench_driver","bench_default"):
    try:
        d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1])
        print(f, "ms/step %.4f"%d["ms_per_step"], "tick_us", d["tick_us"], "kernels", d["kernels_us"], "roof", d["roofline"]["frac"], d["roofline"].get("valu",{}).get("frac"), "f64", d.get("f64_storage"))
    except Exception as e: print(f, "ERR", e)
PY
timeout 400 bash tools/hang_hunt.sh 1200 4 gpurun_out/a/hang 2>&1 | tail -8
