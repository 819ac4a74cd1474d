#!/bin/bash
# Run ON THE GPU BOX (gpurun): the round's full measurement set -> gpurun_out/final/ (what profiles/rN_* is copied from).
#   tests, the bench line (default and driver-like flags), rocprofv3 kernel stats of the same command, the PMC passes,
#   the other BASELINE configs, per-shard sizes (what N = 2/4/8 ranks each run), a fresh-process hang hunt.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -q 2>&1 ) > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log
timeout 400 python bench.py 2>$O/bench_c4.err | tail -1 > $O/bench_c4.json; cp $R/gpurun_out/bench_full_c4.json $O/bench_c4_full_record.json
timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_c4_driver_flags.json
cd /tmp && export TMPDIR=/tmp
# kernel stats: three runs with ONE engine (every rollout launch covers all 10^6 samples: its average is the duration the SURVEY 8(d)
# accounting divides 600 MB by), one run of the default command (the co-scheduled headline: launches of 581 632 and 418 368 samples
# next to each other, plus the one-engine leg's full-size ones -- a mixed average)
for i in 1 2 3; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats$i -- python $R/bench.py --co-shards 1 --no-cpu-baseline --no-f64-line > $O/stats$i.log 2>&1
  cp $O/stats$i/*/*kernel_stats.csv $O/kernel_stats_one_engine_run$i.csv 2>/dev/null
  grep '^{"metric"' $O/stats$i.log | tail -1 > $O/bench_under_rocprof_one_engine_run$i.json
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_co -- python $R/bench.py --no-cpu-baseline --no-f64-line > $O/stats_co.log 2>&1
cp $O/stats_co/*/*kernel_stats.csv $O/kernel_stats_co_headline.csv 2>/dev/null
# config 5 on its two agent-split engines (kernels of two streams next to each other), and on one engine
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -- python $R/bench.py --workload c5 --no-cpu-baseline > $O/stats_c5.log 2>&1
cp $O/stats_c5/*/*kernel_stats.csv $O/kernel_stats_c5_agents_split.csv 2>/dev/null
timeout 300 python $R/bench.py --workload c5 --co-shards 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c5_one_engine.json
timeout 300 python $R/bench.py --co-shards 1 --no-cpu-baseline --no-f64-line 2>/dev/null | tail -1 > $O/bench_c4_one_engine.json
cd $R
timeout 900 bash tools/pmc.sh final/pmc --co-shards 1 > $O/pmc.log 2>&1; tail -4 $O/pmc.log
for w in c2 c3 c5; do timeout 300 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$w.json; done
# one rank's share of config 4 at N = 2 / 4 / 8, as a rank of `bench.py --gpus N` runs it (kernels picked by the whole controller's size:
# --samples-total) and with each share choosing for itself (--rank-kernels)
for k in 500000 250000 125000; do
  timeout 200 python bench.py --samples $k --samples-total 1000000 --co-shards 1 --no-cpu-baseline --no-f64-line --steps 100 2>/dev/null | tail -1 > $O/bench_shard_$k.json
  timeout 200 python bench.py --samples $k --rank-kernels --co-shards 1 --no-cpu-baseline --no-f64-line --steps 100 2>/dev/null | tail -1 > $O/bench_shard_${k}_own_kernels.json
done
python tools/ab_rollout.py --rounds 2 > $O/ab_rollout_c4.jsonl 2>/dev/null
timeout 60 ./tools/depbench > $O/depbench.txt 2>&1
timeout 120 ./tools/rampbench > $O/rampbench.txt 2>&1
timeout 200 python tools/probe_timeline.py --samples 125000,250000,1000000 > $O/probe_timeline.jsonl 2>/dev/null
timeout 120 python tools/probe_timeline.py --samples 100000 --horizon 100 >> $O/probe_timeline.jsonl 2>/dev/null
cp $R/gpurun_out/bench_full_*.json $O/ 2>/dev/null
g++ -O2 -std=c++17 -Iinclude tools/node_tail.cpp -o tools/node_tail -Lmotion_planning_amd/lib -lmppi_hip -Wl,-rpath,$R/motion_planning_amd/lib && ( ./tools/node_tail 10 100 5000 0; ./tools/node_tail 10 100 3000 500; ./tools/node_tail 1000 50 3000 0; ./tools/node_tail 10000 50 3000 0 ) > $O/node_tail.txt 2>&1
( ./tools/node_tail 100000 100 2000 0; ./tools/node_tail 125000 50 2000 0; ./tools/node_tail 1000000 50 1000 0 ) > $O/node_tail_large_k.txt 2>&1
timeout 120 python tools/node_latency.py > $O/node_latency.txt 2>&1
timeout 60 ./tools/ubench > $O/ubench.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/bwbench.hip -o tools/bwbench 2>/dev/null && timeout 120 ./tools/bwbench > $O/bwbench.txt 2>&1
timeout 200 python bench.py --storage f64 --no-cpu-baseline --steps 100 2>/dev/null | tail -1 > $O/bench_c4_f64.json
timeout 200 python bench.py --samples 125000 --samples-total 1000000 --storage f64 --co-shards 1 --no-cpu-baseline --no-f64-line --steps 100 2>/dev/null | tail -1 > $O/bench_shard_125000_f64.json
timeout 300 python tools/ab_fused_f64.py 2>/dev/null | grep "^K" > $O/ab_fused_f64.txt
timeout 120 ./tools/shared_line_repro 3000 200 > $O/shared_line_repro.txt 2>&1
( for co in 0 1 0 1; do ./tools/node_tail 1000000 50 1500 0 $co | head -1; done ) > $O/node_tail_c4_blocking_co_ab.txt 2>&1
# the fused fp64 tick under rocprofv3 and the counters
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_f64 -- python $R/bench.py --storage f64 --no-cpu-baseline --steps 100 > $O/stats_f64.log 2>&1
cp $O/stats_f64/*/*kernel_stats.csv $O/kernel_stats_c4_f64.csv 2>/dev/null
cd $R
timeout 600 bash tools/pmc.sh final/pmc_f64 --storage f64 > $O/pmc_f64.log 2>&1; tail -3 $O/pmc_f64.log
# the driver's N > 1 command line and the plain-process form of it, all ranks on this box's one GPU
timeout 300 python bench.py --gpus 2 --all-ranks-on-gpu0 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c4_two_ranks_one_gpu_self_launched.json
timeout 300 python bench.py --gpus 8 --all-ranks-on-gpu0 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c4_eight_ranks_one_gpu_self_launched.json
[ -z "$RUN_HANG_HUNT" ] || timeout 300 bash tools/hang_hunt.sh 600 4 gpurun_out/final/hang 2>&1 | tail -5
python3 - <<PY
import csv, glob, json, collections
rows = collections.defaultdict(list)
for i in (1, 2, 3):
    for f in glob.glob("$O/stats%d/*/*kernel_stats.csv" % i):
        for r in csv.DictReader(open(f)):
            rows[r["Name"].split("(")[0].replace("void ", "")].append((i, int(r["Calls"]), float(r["AverageNs"]) / 1e3))
json.dump({k: [{"run": a, "calls": b, "avg_us": c} for a, b, c in v] for k, v in rows.items() if "mppi::" in k},
          open("$O/kernel_stats_all_runs.json", "w"), indent=1)
for k, v in rows.items():
    if "mppi::" in k: print(k[:60], [(b, round(c, 2)) for a, b, c in v])
PY
ls $O
