#!/bin/bash
mkdir -p gpurun_out/g2
export TMPDIR=/tmp
O=gpurun_out/g2/ab.jsonl
: > $O
python tools/ab_rollout.py --rounds 2 >> $O 2>> gpurun_out/g2/ab.err
MPPI_PK_WAVES=5 python tools/ab_rollout.py --rounds 2 >> $O 2>> gpurun_out/g2/ab.err
for K in 750000 500000 375000 250000; do python tools/ab_rollout.py --rounds 1 --samples $K >> $O 2>> gpurun_out/g2/ab.err; done
MPPI_PK_WAVES=5 python tools/ab_rollout.py --rounds 1 --samples 500000 >> $O 2>> gpurun_out/g2/ab.err
MPPI_PK_WAVES=5 python tools/ab_rollout.py --rounds 1 --samples 250000 >> $O 2>> gpurun_out/g2/ab.err
python tools/ab_rollout.py --rounds 1 --parked >> $O 2>> gpurun_out/g2/ab.err
cut -c1-330 $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k 'full_size or config5 or f32_storage or many_weighted' > gpurun_out/g2/pytest_subset.log 2>&1
grep -h 'full-size replay\|config 5\|f32 vs f64\|many weighted\|passed\|failed' gpurun_out/g2/pytest_subset.log
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/g2/pytest_all.log 2>&1
tail -15 gpurun_out/g2/pytest_all.log
