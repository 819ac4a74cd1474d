#!/bin/bash
mkdir -p gpurun_out/g4
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_p2p_gpu.py tests/test_gpu_parity.py -m gpu -q -x -k "co_scheduled or many_weighted or update_properties or floor_term or full_size" > gpurun_out/g4/pytest_sub.log 2>&1
tail -8 gpurun_out/g4/pytest_sub.log
python tools/ab_rollout.py --rounds 1 --parked > gpurun_out/g4/ab_parked.jsonl 2> gpurun_out/g4/ab.err
python tools/ab_rollout.py --rounds 1 >> gpurun_out/g4/ab_parked.jsonl 2>> gpurun_out/g4/ab.err
python tools/ab_rollout.py --rounds 1 --samples 125000 >> gpurun_out/g4/ab_parked.jsonl 2>> gpurun_out/g4/ab.err
python - <<'PY'
import json
for l in open('gpurun_out/g4/ab_parked.jsonl'):
    d=json.loads(l); print(d['pk'], d['K'], 'parked' if d['parked'] else 'underway', 'tick %.1f'%d['tick_us'], {k: round(v,1) for k,v in d['bracketed_us'].items()})
PY
timeout 600 python bench.py --steps 100 > gpurun_out/g4/bench_c4.json 2> gpurun_out/g4/bench_c4.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/g4/bench_c4.json').read().strip().splitlines()[-1])
print("value %.4g ms %.4f" % (d['value'], d['ms_per_step'])); print("parked", d.get('parked_at_goal')); o=d['one_engine']; print("one", o['ms_per_step'], o['kernels_us_bracketed'], o['parked_at_goal']); print(d['sync_tick_us'])
PY
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/g4/pytest_all.log 2>&1
tail -8 gpurun_out/g4/pytest_all.log
