"""Step-by-step check of the p2p exchange on one GPU (prints what fails and why)."""
import multiprocessing as mp
import os
import sys
import traceback

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
K, T = 4096, 50
U0 = np.array([np.linspace(-2, 1, T), np.linspace(1.5, -1, T)])


def step(name, fn):
    try:
        r = fn()
        print("[ok]  ", name, r if r is not None else "")
        return True
    except Exception as e:
        print("[FAIL]", name, "->", repr(e))
        traceback.print_exc()
        return False


def single_rank():
    from motion_planning_amd.mppi import Engine
    with Engine(K, T, tick_path="lanes") as a, Engine(K, T, tick_path="lanes") as b:
        for e in (a, b):
            e.set_nominal(U0)
        a.tick_begin([0, 0, 0], [0, -1, 0], noise="philox", seed=1, tick_id=0); a.tick_finish()
        ra = a.get_outputs()
        b.p2p_create(1, 0); b.p2p_connect(local_ptrs=[b.p2p_mailbox_ptr()])
        b.p2p_selftest(3)
        b.tick_begin([0, 0, 0], [0, -1, 0], noise="philox", seed=1, tick_id=0); b.tick_exchange_p2p()
        rb = b.get_outputs()
        assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1]), (ra, rb)
        return "1-rank exchange == tick_finish"


def two_in_process():
    from motion_planning_amd.mppi import Engine
    engs = [Engine(K // 2, T, sample_offset=g * (K // 2), tick_path="lanes") for g in range(2)]
    for g, e in enumerate(engs):
        e.set_nominal(U0); e.p2p_create(2, g)
    ptrs = [e.p2p_mailbox_ptr() for e in engs]
    for e in engs:
        e.p2p_connect(local_ptrs=ptrs)
    for i in range(3):
        for e in engs:
            e.tick_begin([0, 0, 0] if i == 0 else None, [0, -1, 0] if i == 0 else None, noise="philox", seed=1, tick_id=i)
        for e in engs:
            e.p2p_publish()
        for e in engs:
            e.tick_finish_p2p()
        outs = [e.get_outputs() for e in engs]
        assert np.array_equal(outs[0][0], outs[1][0])
    for e in engs:
        e.close()
    return "2 engines agree"


def _ipc_worker(rank, conn):
    try:
        from motion_planning_amd.mppi import Engine
        e = Engine(K // 2, T, sample_offset=rank * (K // 2), tick_path="lanes")
        e.set_nominal(U0)
        h = e.p2p_create(2, rank)
        conn.send(h)
        handles = conn.recv()
        e.p2p_connect(handles=handles)
        conn.send("connected"); conn.recv()
        e.p2p_selftest(4)
        conn.send("selftest ok"); conn.recv()
        outs = []
        for i in range(3):
            e.tick_begin([0, 0, 0] if i == 0 else None, [0, -1, 0] if i == 0 else None, noise="philox", seed=1, tick_id=i)
            e.tick_exchange_p2p()
            outs.append(e.get_outputs()[0].tolist())
        conn.send(outs); conn.recv()
        e.close()
    except Exception as ex:
        conn.send("ERR %r" % (ex,))


def two_processes():
    ctx = mp.get_context("spawn")
    pipes = [ctx.Pipe() for _ in range(2)]
    procs = [ctx.Process(target=_ipc_worker, args=(r, pipes[r][1])) for r in range(2)]
    for p in procs:
        p.start()
    def gather():
        out = []
        for r in range(2):
            if not pipes[r][0].poll(60):
                raise RuntimeError("rank %d silent" % r)
            v = pipes[r][0].recv()
            if isinstance(v, str) and v.startswith("ERR"):
                raise RuntimeError("rank %d: %s" % (r, v))
            out.append(v)
        return out
    handles = gather()
    for r in range(2):
        pipes[r][0].send(handles)
    print("   ", gather())
    for r in range(2): pipes[r][0].send("go")
    print("   ", gather())
    for r in range(2): pipes[r][0].send("go")
    outs = gather()
    for r in range(2): pipes[r][0].send("bye")
    for p in procs:
        p.join(30)
    assert outs[0] == outs[1], outs
    return "2 processes agree: %s" % (outs[0][-1],)


if __name__ == "__main__":
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    step("single rank", single_rank)
    step("two engines, one process", two_in_process)
    step("two processes over IPC", two_processes)
