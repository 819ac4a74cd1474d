#!/bin/bash
mkdir -p gpurun_out/g13
timeout 600 python -m pytest tests/test_p2p_gpu.py -m gpu -q -k "parameter_changes or behind_one_handle or bit_identical" > gpurun_out/g13/pytest_co.log 2>&1
tail -6 gpurun_out/g13/pytest_co.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/g13/bench.json 2> gpurun_out/g13/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/g13/bench.json').read().strip().splitlines()[-1])
r=d['roofline']; print("value %.4g ms %.4f traffic %s one.traffic %s" % (d['value'], d['ms_per_step'], r['traffic'], d['one_engine']['roofline'].get('traffic')))
print(r.get('traffic_source')); print(d['value_parked_at_goal'], d['dtype'])
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
