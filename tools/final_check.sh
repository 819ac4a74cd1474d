#!/bin/bash
# Run ON THE GPU BOX: the last check of a round on one box -- the GPU suite, the bench lines that get committed, three rocprofv3
# kernel-stats runs of the one-engine command and one of the default command -> gpurun_out/lines/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lines; mkdir -p $O; cd $R
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 ) > $O/gpu_tests.log; tail -3 $O/gpu_tests.log
timeout 400 python bench.py 2>$O/bench_c4.err | tail -1 > $O/bench_c4.json
timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_c4_driver_flags.json
timeout 300 python bench.py --co-shards 1 --no-cpu-baseline --no-f64-line 2>/dev/null | tail -1 > $O/bench_c4_one_engine.json
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats$i -- python $R/bench.py --co-shards 1 --no-cpu-baseline --no-f64-line > $O/stats$i.log 2>&1
  cp $O/stats$i/*/*kernel_stats.csv $O/kernel_stats_one_engine_run$i.csv 2>/dev/null
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_co -- python $R/bench.py --no-cpu-baseline --no-f64-line > $O/stats_co.log 2>&1
cp $O/stats_co/*/*kernel_stats.csv $O/kernel_stats_co_headline.csv 2>/dev/null
cd $R
python - <<'PY'
import json, csv, glob
d=json.loads(open('gpurun_out/lines/bench_c4.json').read().strip().splitlines()[-1])
r=d['roofline']; print("value %.4g ms %.4f | roofline %s frac %.3f avg %.1f us valu %.3f clock %.0f | co launch %s" % (d['value'], d['ms_per_step'], r['kernel'], r['frac'], r['avg_launch_us'], r['valu']['frac'], r['valu']['clock_mhz_under_load'], {k: r['co_scheduled_launch'][k] for k in ('avg_launch_us','frac','valu_frac')}))
print("one_engine ms %.4f self_check %s parked %s" % (d['one_engine']['ms_per_step'], d['one_engine']['self_check']['max_abs_diff_u'], d['value_parked_at_goal']))
for i in (1,2,3):
    for f in glob.glob('gpurun_out/lines/kernel_stats_one_engine_run%d.csv' % i):
        for row in csv.DictReader(open(f)):
            if 'rollout' in row['Name']: print(i, row['Name'][:40], row['Calls'], float(row['AverageNs'])/1e3)
PY
