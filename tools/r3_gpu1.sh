#!/bin/bash
# round 3, first GPU call: A/B of the rollout kernels, parity of the new one, ubench with the measured clock
mkdir -p gpurun_out/g1
export TMPDIR=/tmp
python tools/ab_rollout.py --rounds 3 > gpurun_out/g1/ab_c4.jsonl 2> gpurun_out/g1/ab_c4.err
python tools/ab_rollout.py --rounds 1 --samples 125000 >> gpurun_out/g1/ab_c4.jsonl 2>> gpurun_out/g1/ab_c4.err
python tools/ab_rollout.py --rounds 1 --samples 100000 --horizon 100 >> gpurun_out/g1/ab_c4.jsonl 2>> gpurun_out/g1/ab_c4.err
./tools/ubench > gpurun_out/g1/ubench.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_parity.py::test_cpp_node_matches_the_python_shim -k "full_size or config5 or many_weighted or philox or tail_length or f32_storage or smoke" > gpurun_out/g1/pytest_subset.log 2>&1
tail -5 gpurun_out/g1/pytest_subset.log
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/g1/pytest_all.log 2>&1
tail -30 gpurun_out/g1/pytest_all.log
cat gpurun_out/g1/ab_c4.jsonl | cut -c1-400
