# scan kernel vs lane kernels over K (config 2 geometry, T = 50) + the node's blocking call
for K in 1000 4000 10000 16000; do
for M in scan lanes; do
timeout 100 python bench.py --workload c2 --samples $K --tick-path $M --no-cpu-baseline --steps 300 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('K=$K path=$M tick_us=%.1f'%(d['ms_per_step']*1e3), {k:(round(v,1) if v else v) for k,v in d['kernels_us'].items()}, 'blocking', round(d['sync_tick_us']['median'],1))"
done; done
timeout 60 python tools/node_latency.py
