#!/bin/bash
mkdir -p gpurun_out/g3
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_p2p_gpu.py -m gpu -q -x -k "co_scheduled" > gpurun_out/g3/pytest_co.log 2>&1
tail -15 gpurun_out/g3/pytest_co.log
timeout 600 python bench.py --steps 100 > gpurun_out/g3/bench_c4.json 2> gpurun_out/g3/bench_c4.err
tail -3 gpurun_out/g3/bench_c4.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/g3/bench_c4.json').read().strip().splitlines()[-1])
    print("value %.4g ms %.4f co %s" % (d['value'], d['ms_per_step'], d['config'].get('co_samples')))
    r=d['roofline']; print("roofline", r['kernel'], r['frac'], r['avg_launch_us'], r.get('valu',{}).get('frac'), r.get('valu',{}).get('clock_mhz_under_load'), r['tick_level'])
    o=d.get('one_engine'); print("one_engine", o and {k:o[k] for k in ('ms_per_step','rollout_us','shader_clock_mhz','self_check','kernels_us_bracketed')}, o and o['roofline']['frac'], o and o['roofline'].get('valu',{}).get('frac'), o and o['parked_at_goal'])
    print("parked", d.get('parked_at_goal')); print("f64", d.get('f64_storage')); print("tick_us", d['tick_us'], d['sync_tick_us'])
except Exception as e: print("bench parse failed", e)
PY
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/g3/pytest_all.log 2>&1
tail -15 gpurun_out/g3/pytest_all.log
