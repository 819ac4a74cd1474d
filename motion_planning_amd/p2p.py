"""Set-up of the engine's one-shot peer-to-peer all-gather (include/mppi_hip.h mppi_p2p_*) across the ranks of a
torch.distributed group on ONE node.

    setup(engine, group, rank, world, local_rank) -> True when every rank has switched to the p2p exchange

Steps, all collective over the group (object collectives only: works on "nccl" and on "gloo"):
  1. probe (optional, default on real multi-GPU runs): every rank starts a CHILD process that builds a tiny engine
     on the same GPU, exchanges mailbox handles through the parents and runs the self-test.  A GPU memory fault on
     a machine whose peer mappings do not work kills the child, not the benchmark; any rank's failure -> RCCL.
  2. the real engines: mppi_p2p_create -> all-gather of the IPC handles -> mppi_p2p_connect -> mppi_p2p_selftest;
     again any failure on any rank -> every rank tears the mailboxes down and stays on RCCL.
"""
import os
import subprocess
import sys


def _all_gather(dist, group, world, obj):
    out = [None] * world
    dist.all_gather_object(out, obj, group=group)
    return out


def _probe(dist, group, rank, world, local_rank, timeout_s=60.0):
    """Run the handle exchange + self-test in child processes; True when every rank's child passed."""
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    child = subprocess.Popen([sys.executable, "-m", "motion_planning_amd.p2p_probe", str(world), str(rank), str(local_rank)],
                             cwd=here, env=env, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    ok, handle = True, ""
    try:
        import select
        ready, _, _ = select.select([child.stdout], [], [], timeout_s)   # a child that never gets that far is a failed probe
        line = child.stdout.readline().strip() if ready else ""          # "HANDLE <hex>"
        ok = line.startswith("HANDLE ")
        handle = line[7:] if ok else ""
    except Exception:
        ok = False
    handles = _all_gather(dist, group, world, handle if ok else None)
    if all(h for h in handles):
        try:
            child.stdin.write(" ".join(handles) + "\n")
            child.stdin.flush()
            out, _ = child.communicate(timeout=timeout_s)
            ok = child.returncode == 0 and "P2P_OK" in out
        except Exception:
            ok = False
    else:
        ok = False
    if child.poll() is None:
        child.kill()
    try:
        child.wait(timeout=10)
    except Exception:
        pass
    return all(_all_gather(dist, group, world, bool(ok)))


def setup(engine, group, rank, world, local_rank, required=False, probe=None, selftest_rounds=8, report=None):
    """Returns True when every rank has switched to the p2p exchange.  `report` (a dict, optional) is filled with what
    happened on THIS rank, step by step -- bench.py prints it per rank, so that a reader of one JSON line can tell why a
    rank fell back to RCCL: {"probe", "create", "connect", "selftest"} -> "ok" | "skipped" | "failed: <reason>" | "not reached",
    "selftest_round_trip_us", "all_ranks_ok", "peer_reasons"."""
    import time
    import torch.distributed as dist
    rep = report if report is not None else {}
    rep.update({"probe": "not reached", "create": "not reached", "connect": "not reached", "selftest": "not reached",
                "selftest_round_trip_us": None, "all_ranks_ok": False, "peer_reasons": None})
    if world < 2 or world > 8:
        rep["probe"] = "failed: the p2p exchange serves 2..8 ranks on one node (world = %d)" % world
        if required:
            raise RuntimeError("the p2p exchange serves 2..8 ranks on one node")
        return False
    if probe is None:
        probe = dist.get_backend(group) == "nccl"     # real multi-GPU run: guard the first contact
    if probe:
        ok_probe = _probe(dist, group, rank, world, local_rank)
        rep["probe"] = "ok" if ok_probe else "failed: a rank's child process did not complete the handle exchange + self-test"
        if not ok_probe:
            if required:
                raise RuntimeError("p2p probe failed on at least one rank")
            return False
    else:
        rep["probe"] = "skipped"
    ok, handle, why = True, None, []
    try:
        handle = engine.p2p_create(world, rank)
        rep["create"] = "ok"
    except Exception as e:
        ok = False
        why.append("create: %s" % e)
        rep["create"] = "failed: %s" % e
    handles = _all_gather(dist, group, world, handle)
    ok = ok and all(h is not None for h in handles)
    if ok:
        try:
            engine.p2p_connect(handles=handles)
            rep["connect"] = "ok"
        except Exception as e:
            ok = False
            why.append("connect: %s" % e)
            rep["connect"] = "failed: %s" % e
    elif rep["create"] == "ok":
        rep["connect"] = "not reached: a peer has no mailbox"
    ok = all(_all_gather(dist, group, world, ok))
    if ok:
        try:
            t0 = time.perf_counter()
            engine.p2p_selftest(selftest_rounds)      # collective: every rank publishes to every rank
            rep["selftest"] = "ok"
            rep["selftest_round_trip_us"] = 1e6 * (time.perf_counter() - t0) / max(selftest_rounds, 1)
        except Exception as e:
            ok = False
            why.append("selftest: %s" % e)
            rep["selftest"] = "failed: %s" % e
        ok = all(_all_gather(dist, group, world, ok))
    elif rep["connect"] == "ok":
        rep["selftest"] = "not reached: a peer failed to connect"
    rep["all_ranks_ok"] = bool(ok)
    if not ok:
        try:
            engine.p2p_destroy()
        except Exception:
            pass
        reasons = _all_gather(dist, group, world, "; ".join(why))
        rep["peer_reasons"] = reasons
        if required:
            raise RuntimeError("p2p exchange could not be established on every rank: %s" % reasons)
        if rank == 0 and os.environ.get("MPPI_P2P_VERBOSE"):
            sys.stderr.write("p2p exchange not used: %s\n" % reasons)
    return ok
