#!/usr/bin/env python3
"""Same-box A/B of the fused tick (tick_fused_kernel: rollout + update work items of one launch) against the stand-alone
kernels, through the product's own Engine and mppi_set_option switches.

    python tools/fused_ab.py [--quick] [--out gpurun_out/fused_ab.jsonl]

Per configuration: closed-loop device-noise ticks, time-based warm-up, then `--steps` back-to-back ticks between two
synchronisations; every variant must end in the same state / controls as the stand-alone run (1e-10), and at the small
sizes V is compared bit for bit.  One JSON line per (configuration, variant)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from motion_planning_amd.mppi import Engine  # noqa: E402


def nominal_warm(T):
    return np.array([np.linspace(-2.0, 1.0, T), np.linspace(1.5, -1.0, T)])


def run(K, T, A, co, opts, steps, want_v=False, kernels=("fused", "rollout", "update", "merge", "finalize")):
    goal = np.tile([0.0, -1.0, 0.0], (A, 1))
    with Engine(K, T, n_agents=A, storage="f32", tick_path="lanes", co_shards=co, options=opts) as e:
        for a in range(A):
            e.set_nominal(nominal_warm(T), agent=a)
        e.tick_async(np.zeros((A, 3)), goal, "philox", 0, 0)
        t_w, i = time.perf_counter(), 1
        while time.perf_counter() - t_w < 0.3 or i < 20:
            e.tick_async(None, None, "philox", 0, i)
            i += 1
            if i % 16 == 0:
                e.synchronize()
        for a in range(A):
            e.set_nominal(nominal_warm(T), agent=a)
        e.tick_async(np.zeros((A, 3)), goal, "philox", 0, 1_000_000)
        e.synchronize()
        t0 = time.perf_counter()
        for j in range(steps):
            e.tick_async(None, None, "philox", 0, 1_000_001 + j)
        e.synchronize()
        el = time.perf_counter() - t0
        nxt, ua = e.get_outputs()
        info = e.info()
        e.kernel_timing(kernels, period=1)
        for j in range(20):
            e.tick_async(None, None, "philox", 0, 3_000_000 + j)
        e.synchronize()
        kt = e.kernel_times()
        e.kernel_timing(())
        V = None
        if want_v:
            for a in range(A):
                e.set_nominal(nominal_warm(T), agent=a)
            e.tick(np.zeros((A, 3)), goal, noise="philox", seed=3, tick_id=77)
            V = e.download_value()
    return {"tick_us": 1e6 * el / steps, "state": nxt.copy(), "u": ua.copy(), "fused": info["tick_fused"], "rollout_kernel": info["rollout_kernel"],
            "co": info["co_shards"], "kernels_us": {k: (v[0] * 1e3 / v[1] if v[1] else None) for k, v in kt.items()}, "V": V}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--out", default=None)
    ap.add_argument("--configs", default="c4,share125k,share250k,share500k,c5,small")
    ap.add_argument("--family", default="stream", choices=["one", "stream", "both"], help="one: the single fused launch; stream: the update stream")
    args = ap.parse_args()
    configs = {
        "c4": (1000000, 50, 1), "share500k": (500000, 50, 1), "share250k": (250000, 50, 1), "share125k": (125000, 50, 1),
        "c5": (16384, 50, 64), "small": (40000, 50, 1), "c3": (100000, 100, 1),
    }
    variants = [("standalone", {"fused": 0})]
    if args.family in ("one", "both"):
        variants += [("fused", {"fused": 1}), ("fused_p1", {"fused": 1, "fused_prio": 1}), ("fused_p2", {"fused": 1, "fused_prio": 2}),
                     ("fused_p3", {"fused": 1, "fused_prio": 3})]
        if not args.quick:
            variants += [("fused_p3_lag2", {"fused": 1, "fused_prio": 3, "fused_lag": 2}), ("fused_p3_lag4", {"fused": 1, "fused_prio": 3, "fused_lag": 4}),
                         ("fused_lag4", {"fused": 1, "fused_lag": 4}), ("fused_lag16", {"fused": 1, "fused_lag": 16}),
                         ("fused_p1_lag4", {"fused": 1, "fused_prio": 1, "fused_lag": 4})]
    if args.family in ("stream", "both"):
        variants += [("us", {"fused": 2}), ("us_p1", {"fused": 2, "fused_prio": 1}), ("us_p2", {"fused": 2, "fused_prio": 2}), ("us_p3", {"fused": 2, "fused_prio": 3})]
        if not args.quick:
            variants += [("us_t512", {"fused": 2, "us_tail_blocks": 512}), ("us_t1024", {"fused": 2, "us_tail_blocks": 1024}),
                         ("us_p2_t1024", {"fused": 2, "fused_prio": 2, "us_tail_blocks": 1024}), ("us_p3_t1024", {"fused": 2, "fused_prio": 3, "us_tail_blocks": 1024}),
                         ("us_b512_t1024", {"fused": 2, "us_blocks": 512, "us_tail_blocks": 1024}), ("us_b128_t1024", {"fused": 2, "us_blocks": 128, "us_tail_blocks": 1024}),
                         ("us_b64_t1024", {"fused": 2, "us_blocks": 64, "us_tail_blocks": 1024})]
    out = open(args.out, "w") if args.out else None
    bad = 0
    for name in args.configs.split(","):
        K, T, A = configs[name]
        want_v = A * K * T <= 12_500_000
        ref = None
        for co in ((1, 2) if name == "c4" else (1,)):
            for vname, opts in variants:
                if co == 2 and vname not in ("standalone", "fused", "fused_p3", "us", "us_p3"):
                    continue
                r = run(K, T, A, co, opts, args.steps, want_v=want_v)
                if ref is None:
                    ref = r
                ds = float(np.abs(r["state"] - ref["state"]).max())
                du = float(np.abs(r["u"] - ref["u"]).max())
                v_same = None if r["V"] is None else bool(np.array_equal(r["V"], ref["V"]))
                ok = ds <= 1e-10 and du <= 1e-10 and v_same is not False and r["fused"] == opts.get("fused", 0)
                bad += 0 if ok else 1
                line = {"config": name, "K": K, "T": T, "A": A, "co_shards": r["co"], "variant": vname, "options": opts, "tick_us": r["tick_us"],
                        "fused_ran": r["fused"], "rollout_kernel": r["rollout_kernel"], "kernels_us": r["kernels_us"],
                        "max_abs_diff_state": ds, "max_abs_diff_u": du, "V_bit_identical": v_same, "ok": ok}
                print(json.dumps(line), flush=True)
                if out:
                    out.write(json.dumps(line) + "\n")
                    out.flush()
    if bad:
        raise SystemExit("%d variant(s) differ from the stand-alone kernels" % bad)


if __name__ == "__main__":
    main()
