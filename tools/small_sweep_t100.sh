# the same at T = 100 (a block of four waves per sample on the scan path)
for K in 500 2000 5000 10000; do
for M in scan lanes; do
timeout 100 python bench.py --workload c2 --horizon 100 --samples $K --tick-path $M --no-cpu-baseline --steps 300 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('T=100 K=$K path=$M tick_us=%.1f'%(d['ms_per_step']*1e3), {k:(round(v,1) if v else v) for k,v in d['kernels_us'].items()}, 'blocking', round(d['sync_tick_us']['median'],1))"
done; done
