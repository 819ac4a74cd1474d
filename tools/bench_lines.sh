#!/bin/bash
# Run ON THE GPU BOX: the bench lines that get committed under profiles/ (default flags, the driver's flags, the other
# BASELINE configs) + three rocprofv3 kernel-stats runs of the default command -> gpurun_out/lines/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lines; mkdir -p $O; cd $R
timeout 400 python bench.py 2>/dev/null | tail -1 > $O/bench_c4.json
timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_c4_driver_flags.json
for w in c2 c3 c5; do timeout 300 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$w.json; done
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats$i -- python $R/bench.py --no-cpu-baseline --no-f64-line --no-co-line > $O/stats$i.log 2>&1
  cp $O/stats$i/*/*kernel_stats.csv $O/kernel_stats_run$i.csv 2>/dev/null
  tail -1 $O/stats$i.log > $O/bench_under_rocprof_run$i.json
done
cd $R; ls $O; head -c 400 $O/bench_c4.json
