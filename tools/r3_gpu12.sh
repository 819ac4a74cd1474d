#!/bin/bash
mkdir -p gpurun_out/g12
export TMPDIR=/tmp
O=gpurun_out/g12/ab.jsonl; : > $O
MPPI_PK_AHEAD=0 python tools/ab_rollout.py --rounds 1 >> $O 2>> gpurun_out/g12/ab.err
MPPI_PK_AHEAD=1 python tools/ab_rollout.py --rounds 1 >> $O 2>> gpurun_out/g12/ab.err
MPPI_PK_AHEAD=0 python tools/ab_rollout.py --rounds 1 >> $O 2>> gpurun_out/g12/ab.err
MPPI_PK_AHEAD=1 python tools/ab_rollout.py --rounds 1 >> $O 2>> gpurun_out/g12/ab.err
MPPI_PK_AHEAD=1 MPPI_PK_WAVES=5 python tools/ab_rollout.py --rounds 1 >> $O 2>> gpurun_out/g12/ab.err
python - <<'PY'
import json
for l in open('gpurun_out/g12/ab.jsonl'):
    d=json.loads(l); print(d['pk'], 'tick %.1f rollout %.1f mhz %.0f'%(d['tick_us'],d['rollout_us'],d['shader_mhz']), d['u_applied'])
PY
tail -2 gpurun_out/g12/ab.err
