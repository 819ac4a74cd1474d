set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/final
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --steps 200 --warmup 20 2>&1 | tail -1 > $R/gpurun_out/final/bench_c4.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/stats -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $R/gpurun_out/final/stats.log 2>&1
cd $R
timeout 600 bash tools/pmc.sh final/pmc > $R/gpurun_out/final/pmc.log 2>&1
tail -5 $R/gpurun_out/final/pmc.log
for w in c2 c3 c5; do timeout 200 python bench.py --workload $w --no-cpu-baseline 2>&1 | tail -1 > $R/gpurun_out/final/bench_$w.json; done
ls -R $R/gpurun_out/final | head -40
