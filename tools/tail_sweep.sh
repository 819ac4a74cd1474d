# rollout launch time against K around whole multiples of (1024 SIMDs x 64 lanes): how much is the partial last round?
for K in 917504 983040 1000000 1015808 1048576 1310720; do
python bench.py --samples $K --no-cpu-baseline --no-f64-line --steps 200 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=$K waves/SIMD=%.2f'%($K/65536.0), 'tick_us %.1f'%(d['ms_per_step']*1e3), 'rollout_us %.1f'%d['roofline']['avg_launch_us'], 'per 65536 samples: %.2f us'%(d['roofline']['avg_launch_us']/($K/65536.0)))"
done
