#!/bin/bash
# Fresh-process stress of the C++ node (examples/mppi_node.cpp): does a freshly started control process
# always come back?  Usage (on the GPU box):  tools/hang_hunt.sh [RUNS=1500] [WORKERS=4] [OUTDIR=gpurun_out/hang]
# Every run is a new process doing 30 callbacks (0.3 s).  A run that is still alive after 8 s is a HANG:
# its progress word (MPPI_NODE_TRACE: -1 creating, -2 created, i >= 0 callback i done, -3 destroying,
# -4 destroyed) and a rocgdb backtrace of all threads are saved before it is killed.  The engine's own
# blocking waits are bounded (MPPI_SYNC_TIMEOUT_MS): a device that stops answering shows up as exit code 2
# with the MPPI_E_TIMEOUT message in <OUTDIR>/fail_*.log, not as a hang.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
RUNS=${1:-1500}; WORKERS=${2:-4}; OUT=${3:-gpurun_out/hang}
mkdir -p "$OUT"; make -C examples >/dev/null 2>&1
worker() {
  w=$1; n=$2; hangs=0; fails=0
  for i in $(seq 1 "$n"); do
    P=lanes; [ $((i % 2)) -eq 0 ] && P=scan
    trace=/tmp/progress_$w.bin
    MPPI_SYNC_TIMEOUT_MS=4000 MPPI_NODE_TRACE=$trace ./build/mppi_node --task park --samples 2048 --horizon 50 \
        --callbacks 30 --seed 5 --storage f64 --tick-path $P > /dev/null 2> "$OUT/err_$w.log" &
    pid=$!
    for t in $(seq 1 160); do kill -0 $pid 2>/dev/null || break; sleep 0.05; done
    if kill -0 $pid 2>/dev/null; then
      hangs=$((hangs+1))
      echo "worker $w run $i path=$P HANG progress=$(od -An -i -N4 $trace)" | tee -a "$OUT/hangs.txt"
      timeout 60 /opt/rocm/bin/rocgdb -p $pid -batch -ex "thread apply all bt" > "$OUT/bt_${w}_$i.txt" 2>&1
      kill -9 $pid 2>/dev/null; wait $pid 2>/dev/null
    else
      wait $pid; rc=$?
      if [ $rc -ne 0 ]; then
        fails=$((fails+1)); cp "$OUT/err_$w.log" "$OUT/fail_${w}_$i.log"
        echo "worker $w run $i path=$P rc=$rc progress=$(od -An -i -N4 $trace): $(head -c 200 "$OUT/err_$w.log")" | tee -a "$OUT/hangs.txt"
      fi
    fi
    [ $((hangs + fails)) -ge 4 ] && break
  done
  echo "worker $w: $hangs hangs, $fails failures in $i runs" | tee -a "$OUT/summary.txt"
}
: > "$OUT/summary.txt"; : > "$OUT/hangs.txt"
per=$(( (RUNS + WORKERS - 1) / WORKERS ))
for w in $(seq 1 "$WORKERS"); do worker "$w" "$per" & done
wait
cat "$OUT/summary.txt"
