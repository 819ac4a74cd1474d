#!/bin/bash
mkdir -p gpurun_out/g11
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/g11/pytest_all.log 2>&1
tail -6 gpurun_out/g11/pytest_all.log
