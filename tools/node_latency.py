"""Blocking-call latency of the node's own configurations (what Controller.pos_cb pays per odometry message),
with a breakdown: python shim vs the C call, host I/O (state/goal in, controls out) vs kernels only."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from motion_planning_amd import MPPI
from motion_planning_amd.mppi import Engine


def med(f, n=300, warm=30):
    for _ in range(warm):
        f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(1e6 * (time.perf_counter() - t0))
    ts = np.sort(ts)
    return float(np.median(ts)), float(ts[int(0.99 * n)])


for K, T in [(10, 100), (1000, 50), (10000, 50)]:
    m = MPPI(horizon=T, samples=K, rng="philox")
    st = [np.zeros(3)]; g = np.array([0.0, -1.0, 0.0])
    def f_get_path():
        st[0] = m.get_path(st[0], g)
    a = med(f_get_path)
    e = Engine(K, T)
    s = [np.zeros((1, 3))]; gg = np.array([[0.0, -1.0, 0.0]])
    i = [0]
    def f_tick():
        i[0] += 1; s[0], _ = e.tick(s[0], gg, noise="philox", seed=0, tick_id=i[0])
    def f_resident():   # no host inputs: state and goal stay on the device; outputs still come back
        i[0] += 1; e.tick(None, None, noise="philox", seed=0, tick_id=i[0])
    def f_kernels():    # kernels only: enqueue + wait, no copies in either direction
        i[0] += 1; e.tick_begin(None, None, noise="philox", seed=0, tick_id=i[0]); e.tick_finish(); e.synchronize()
    b, c, d = med(f_tick), med(f_resident), med(f_kernels)
    e.kernel_timing(("nominal", "rollout", "update", "merge", "finalize"))
    for _ in range(50):
        f_kernels()
    kt = e.kernel_times()
    ks = {k: round(v[0] * 1e3 / v[1], 2) for k, v in kt.items() if v[1]}
    print("K=%d T=%d (%s): MPPI.get_path %.1f (p99 %.1f) | Engine.tick %.1f (p99 %.1f) | resident inputs %.1f | kernels only %.1f | kernel us %s"
          % (K, T, e.info()["tick_kernels"], a[0], a[1], b[0], b[1], c[0], d[0], ks))
    e.close()
