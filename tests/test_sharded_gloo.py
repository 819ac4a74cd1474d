"""The N>1 path on CPU: two processes, gloo backend, the product's ShardedTicker driving an
oracle-backed shard (the HIP engine needs a GPU; the exchange logic does not).  Checks that
K split over 2 ranks + one all-gather per tick reproduces the unsharded controller, tick
after tick, on every rank."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIG, LAM = 0.9, 0.001


class OracleShard(object):
    """HipShard's interface on top of the CPU oracle, for one rank's slice [lo, hi) of K."""

    def __init__(self, orc, K_total, lo, hi, T, u0):
        import torch
        self.orc, self.K, self.lo, self.hi, self.T = orc, K_total, lo, hi, T
        self.u = u0.copy()
        self.S = orc.savgol_matrix(T)
        self.part = torch.zeros(T * 8, dtype=torch.float64)
        self.state = None
        self.goal = None
        self.out = None

    def tick_begin(self, state, goal, noise, seed, tick_id):
        import torch
        if state is not None:
            self.state = np.asarray(state, dtype=np.float64).reshape(3)
        if goal is not None:
            self.goal = np.asarray(goal, dtype=np.float64).reshape(3)
        # global sample ids key the stream: this rank draws only its own slice
        eps = self.orc.philox_noise(seed, 0, tick_id, self.lo, self.hi - self.lo, self.T, SIG)
        V = self.orc.get_cost2go(self.state, self.u, self.goal, LAM, SIG, eps)
        p = self.orc.shard_partials(eps, V, 0, self.hi - self.lo, LAM)          # [T][6]
        full = np.zeros((self.T, 8))
        full[:, :6] = p
        full[:, 6] = self.hi - self.lo
        self.part.copy_(torch.from_numpy(full.reshape(-1)))

    def partials_tensor(self):
        return self.part

    def tick_finish(self, gathered, n_shards):
        g = (self.part if gathered is None else gathered).numpy().reshape(n_shards, self.T, 8)
        du = self.orc.merge_partials(np.ascontiguousarray(g[:, :, :6]), g[:, 0, 6], LAM)
        un = np.clip(self.u + du, -6.35492, 6.35492)
        uf = np.clip(un @ self.S, -6.35492, 6.35492)
        nxt = self.orc.rk4(self.state, uf[:, 0], 1.0 / self.T)
        self.out = (nxt[None].copy(), uf[:, 0][None].copy())
        self.state = nxt
        self.u = np.concatenate([uf[:, 1:], np.zeros((2, 1))], axis=1)

    def get_outputs(self):
        return self.out


def _worker(rank, world, port, K, T, n_ticks, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from motion_planning_amd import sharded
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        u0 = np.array([np.linspace(-2, 1, T), np.linspace(1.5, -1, T)])
        lo, hi = sharded.shard_range(K, world, rank)
        ticker = sharded.ShardedTicker(OracleShard(orc, K, lo, hi, T, u0))
        assert ticker.world == world and ticker.rank == rank
        outs = []
        for i in range(n_ticks):
            first = i == 0
            nxt, ua = ticker.tick([0.0, 0.0, 0.0] if first else None, [0.0, -1.0, 0.0] if first else None,
                                  "philox", 7, i)
            outs.append(np.concatenate([nxt[0], ua[0]]))
        q.put((rank, np.array(outs), ticker.shard.u))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_range_is_a_partition():
    from motion_planning_amd.sharded import shard_range
    for K in (1, 7, 64, 1000003):
        for W in (1, 2, 3, 8):
            r = [shard_range(K, W, i) for i in range(W)]
            assert r[0][0] == 0 and r[-1][1] == K
            assert all(r[i][1] == r[i + 1][0] for i in range(W - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("world", [2, 3, 4, 8])   # 2, 4, 8: the world sizes of the driver's scaling runs; 3: ragged thirds
def test_sharded_ticks_equal_unsharded_gloo(orc, world):
    import torch.multiprocessing as mp
    K, T, n_ticks = 301, 20, 3   # K not divisible by the world size: ragged shards
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, K, T, n_ticks, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference: the same controller with all K samples on one shard
    from motion_planning_amd import sharded
    u0 = np.array([np.linspace(-2, 1, T), np.linspace(1.5, -1, T)])
    one = sharded.ShardedTicker(OracleShard(orc, K, 0, K, T, u0))
    ref = []
    for i in range(n_ticks):
        first = i == 0
        nxt, ua = one.tick([0.0, 0.0, 0.0] if first else None, [0.0, -1.0, 0.0] if first else None, "philox", 7, i)
        ref.append(np.concatenate([nxt[0], ua[0]]))
    ref = np.array(ref)
    for rank, outs, u in res:
        assert np.abs(outs - ref).max() < 1e-12, rank
        assert np.abs(u - one.shard.u).max() < 1e-12, rank
    # and the unsharded controller is the oracle's own get_path on the same noise
    st, lat = np.zeros(3), u0.copy()
    for i in range(n_ticks):
        eps = orc.philox_noise(7, 0, i, 0, K, T, SIG)
        st, ua, lat = orc.get_path(st, [0.0, -1.0, 0.0], lat, eps, LAM, SIG)
        assert np.abs(np.concatenate([st, ua]) - ref[i]).max() < 1e-12


class _FakeP2PEngine(object):
    """What p2p.setup needs of an Engine, without a GPU: records the calls, can be told to fail at a step."""

    def __init__(self, rank, fail_at=None):
        self.rank, self.fail_at, self.calls, self._lib = rank, fail_at, [], self

    mppi_p2p_create = True   # `hasattr(lib, "mppi_p2p_create")`

    def p2p_create(self, world, rank):
        self.calls.append("create")
        if self.fail_at == "create":
            raise RuntimeError("no fine-grained memory")
        return bytes([rank]) * 64

    def p2p_connect(self, handles=None, local_ptrs=None):
        self.calls.append("connect")
        assert [h[0] for h in handles] == list(range(len(handles)))      # rank-major, every rank's handle
        if self.fail_at == "connect":
            raise RuntimeError("hipIpcOpenMemHandle failed")

    def p2p_selftest(self, rounds):
        self.calls.append("selftest")
        if self.fail_at == "selftest":
            raise RuntimeError("a flag never arrived")

    def p2p_destroy(self):
        self.calls.append("destroy")


def _p2p_setup_worker(rank, world, port, fail_rank, fail_at, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from motion_planning_amd import p2p
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng = _FakeP2PEngine(rank, fail_at if rank == fail_rank else None)
        rep = {}
        ok = p2p.setup(eng, None, rank, world, 0, required=False, probe=False, report=rep)
        q.put((rank, ok, eng.calls, rep))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fail_at", [None, "create", "connect", "selftest"])
def test_p2p_setup_is_all_or_nothing(fail_at):
    """The decision to leave RCCL for the p2p exchange is collective: one rank failing at any step (mailbox allocation,
    mapping a peer, the self-test) puts EVERY rank back on RCCL with its mailbox torn down; no rank is left waiting in a
    collective the others skipped."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_p2p_setup_worker, args=(r, 2, port, 1, fail_at, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, calls, rep in res:
        assert ok == (fail_at is None), (rank, ok, calls)
        assert rep["all_ranks_ok"] == ok and rep["probe"] == "skipped"
        if fail_at is None:
            assert calls == ["create", "connect", "selftest"]
            assert (rep["create"], rep["connect"], rep["selftest"]) == ("ok", "ok", "ok") and rep["selftest_round_trip_us"] >= 0
        else:
            assert calls[-1] == "destroy"                       # the mailbox is torn down on EVERY rank, also after a failed self-test
            assert fail_at == "selftest" or "selftest" not in calls[:-1]
            # the report says, on the failing rank, which step failed and why; on every rank, what its peers reported
            if rank == 1:
                assert rep[fail_at].startswith("failed: "), rep
            assert any(fail_at in (r or "") for r in rep["peer_reasons"]), rep


def test_co_scheduled_cuts_partition_the_samples():
    from motion_planning_amd.sharded import co_scheduled_cuts
    for K in (2, 3, 9, 6000, 8192, 50000, 300000, 1000000, 1048576, 2**24 + 5):
        for n in (2, 3, 4, 8):
            if K < n:
                with pytest.raises(ValueError):
                    co_scheduled_cuts(K, n)
                continue
            cuts = co_scheduled_cuts(K, n)
            assert cuts[0] == 0 and cuts[-1] == K and len(cuts) == n + 1
            assert all(b > a for a, b in zip(cuts, cuts[1:])), (K, n, cuts)      # every shard has samples
            if K >= 4 * n * 8192:                                                 # room for chunk-aligned boundaries
                assert all(c % 8192 == 0 for c in cuts[1:-1]), (K, n, cuts)
                sizes = [b - a for a, b in zip(cuts, cuts[1:])]
                assert max(sizes) - min(sizes) <= 2 * 8192
    assert co_scheduled_cuts(1000000, 2) == [0, 499712, 1000000]
    with pytest.raises(ValueError):
        co_scheduled_cuts(100, 9)
