"""Run ON THE GPU BOX: the co-scheduled tick against the one-engine tick at sample counts whose per-wave noise-sum rows are / are not
multiples of a 128-byte line (K = 10^6: 15 625 four-byte sums per row; 2^20 and 999 424: multiples of 32).  With MPPI_AB_LIB=aliasep
(make VARIANT=aliasep EXTRA=-DMPPI_ALIAS_EPART_TOO: the second shard's sums as columns of the handle's rows too) the first differs by
2e-7 and the other two are exact: two engines' concurrent kernels must not share a cache line (EXPERIMENTS.md 56)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_amd import _capi
if os.environ.get('MPPI_AB_LIB'): _capi.LIB_PATH = os.path.join(os.path.dirname(_capi.LIB_PATH), 'libmppi_hip_%s.so' % os.environ['MPPI_AB_LIB'])
from motion_planning_amd.mppi import Engine
T = 50
for K in (1000000, 1048576, 999424):
    outs = {}
    for co in (1, 2):
        with Engine(K, T, co_shards=co) as e:
            u0 = np.array([np.linspace(-2, 1, T), np.linspace(1.5, -1, T)])
            e.set_nominal(u0)
            st = np.zeros((1, 3)); goal = np.array([[0.0, -1.0, 0.0]])
            tr = []
            for i in range(6):
                st, ua = e.tick(st if i == 0 else None, goal if i == 0 else None, seed=3, tick_id=i)
                tr.append(np.concatenate([np.asarray(st).ravel(), np.asarray(ua).ravel()]))
            outs[co] = np.array(tr)
    print(os.environ.get('MPPI_AB_LIB', 'default'), K, "NW %% 32 = %d" % ((K + 63) // 64 % 32), "max |co - one| =", float(np.abs(outs[1] - outs[2]).max()), flush=True)
