import sys, time, numpy as np
sys.path.insert(0, '.')
from motion_planning_amd import MPPI
for K, T in [(10, 100), (1000, 50), (10000, 50)]:
    m = MPPI(horizon=T, samples=K, rng="philox")
    st = np.zeros(3); g = np.array([0.0, -1.0, 0.0])
    for _ in range(20): st = m.get_path(st, g)
    t0 = time.perf_counter()
    for _ in range(200): st = m.get_path(st, g)
    print("MPPI.get_path K=%d T=%d: %.1f us/tick" % (K, T, 1e6 * (time.perf_counter() - t0) / 200))
