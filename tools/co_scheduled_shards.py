"""Experiment: G engines on ONE GPU in one process, K split between them, each on its own stream and
coupled only by the p2p mailboxes (flags polled by the finalize kernels) -- against one engine with all K, on the same
box, clocks warm.  One half's HBM-bound update kernel overlaps the other half's VALU-bound rollout.
    python tools/co_scheduled_shards.py [ticks per repetition]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from motion_planning_amd.mppi import Engine

T = 50
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
u0 = np.array([np.linspace(-2, 1, T), np.linspace(1.5, -1, T)])


def single():
    with Engine(K, T, storage="f32") as e:
        e.set_nominal(u0)
        e.tick_async([0, 0, 0], [0, -1, 0], "philox", 0, 0)
        for rep in range(3):
            e.synchronize(); t0 = time.perf_counter()
            for i in range(N):
                e.tick_async(None, None, "philox", 0, 1 + i)
            e.synchronize(); dt = time.perf_counter() - t0
        return 1e6 * dt / N, e.get_outputs()[1][0]


def multi(G, fr=None):
    fr = fr or [1.0 / G] * G
    CH = 8192
    cuts = [0]
    for f in fr[:-1]:
        cuts.append(min(K, int(round((cuts[-1] + f * K) / CH)) * CH))
    cuts.append(K)
    engs = [Engine(cuts[g + 1] - cuts[g], T, storage="f32", sample_offset=cuts[g], tick_path="lanes") for g in range(G)]
    try:
        for g, e in enumerate(engs):
            e.set_nominal(u0); e.p2p_create(G, g)
        ptrs = [e.p2p_mailbox_ptr() for e in engs]
        for e in engs:
            e.p2p_connect(local_ptrs=ptrs)
        def tick(i, first=False):
            for e in engs:
                e.tick_begin([0, 0, 0] if first else None, [0, -1, 0] if first else None, noise="philox", seed=0, tick_id=i)
            for e in engs:
                e.p2p_publish()
            for e in engs:
                e.tick_finish_p2p()
        tick(0, True)
        for rep in range(3):
            for e in engs: e.synchronize()
            t0 = time.perf_counter()
            for i in range(N):
                tick(1 + i)
            for e in engs: e.synchronize()
            dt = time.perf_counter() - t0
        return 1e6 * dt / N, engs[0].get_outputs()[1][0]
    finally:
        for e in engs: e.close()

print("single  %.1f us/tick" % single()[0])
for G, fr in ((2, None), (3, None), (2, None)):
    t, u = multi(G, fr)
    print("G=%d  %.1f us/tick" % (G, t))
print("single  %.1f us/tick" % single()[0])
