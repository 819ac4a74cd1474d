#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/d; mkdir -p $O
cd $R
timeout 300 python tools/p2p_debug.py 2>&1 | tail -40 | cut -c1-300
( time timeout 1200 python -m pytest tests -m gpu -q --timeout 300 -x 2>&1 ) > $O/test.log 2>&1
tail -15 $O/test.log | cut -c1-300
timeout 200 python tools/node_latency.py 2>&1 | tee $O/node_latency.txt
