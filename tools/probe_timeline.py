#!/usr/bin/env python3
"""Timeline of ONE wave inside a rollout launch (the probe wave: thread 0 of the middle workgroup), from the diagnostic build of the
library (make -C motion_planning_amd/csrc PROBE=1 -> lib/libmppi_hip_probe.so): shader cycles behind the prologue's barrier, behind
every chunk of six steps, at the end.  Says where an under-filled launch spends its time: prologue, first chunk (cold instruction
cache), steady-state chunks against their issue cycles.

    python tools/probe_timeline.py [--samples 125000] [--horizon 50] [--storage f32] [--options low_occ=0,table_hoist=0]
One JSON line per configuration."""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from motion_planning_amd import _capi
_LIB = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--lib=")]   # another diagnostic build: --lib=NAME -> lib/libmppi_hip_NAME.so
_capi.LIB_PATH = os.path.join(ROOT, "motion_planning_amd", "lib", "libmppi_hip_%s.so" % (_LIB[0] if _LIB else "probe"))
from motion_planning_amd.mppi import Engine


def run(K, T, storage, opts, ticks=40):
    with Engine(K, T, storage=storage, co_shards=1, tick_path="lanes", options=opts) as e:
        u0 = np.array([np.linspace(-2, 1, T), np.linspace(1.5, -1, T)])
        e.set_nominal(u0)
        e.tick_async(np.zeros((1, 3)), np.array([[0.0, -1.0, 0.0]]), seed=0, tick_id=0)
        for i in range(ticks):
            e.tick_async(seed=0, tick_id=1 + i)
        e.synchronize()
        marks, total = e.probe_timeline()
        mhz = e.shader_clock_mhz()
        e.kernel_timing(("rollout", "update", "finalize"), period=1)
        for i in range(20):
            e.tick_async(seed=0, tick_id=100 + i)
        e.synchronize()
        kt = e.kernel_times()
        kind = e.info()["rollout_kernel"]
    fin = [x for x in marks[16:24] if x]
    upd = [x for x in marks[24:] if x]
    m = [x for x in marks[:16] if x]
    deltas = [m[0]] + [m[i] - m[i - 1] for i in range(1, len(m))]
    return {"K": K, "T": T, "storage": storage, "options": opts, "kernel": kind, "clock_mhz": mhz, "wave_total_cycles": total,
            "wave_total_us": total / mhz if mhz else None, "stamps": m, "deltas_cycles": deltas, "finalize_stamps_cycles": fin, "update_stamps_cycles": upd,
            "update_what": "cycles since the middle update workgroup's first instruction: [chunk loaded, block minimum, weights, re-draws + eps sums, tuple stored]",
            "finalize_what": "cycles since the finalize workgroup's first instruction: [basis loads issued, tuples merged + controls updated, barrier, filter coefficients, filtered controls, stage angles + controls written, plant step + outputs]",
            "what": "deltas: [0] wave start -> behind the prologue's barrier; [1..] each chunk of six steps (the last entries: ride / tail chunk + terminal)",
            "bracketed_us": {k: 1e3 * v[0] / max(v[1], 1) for k, v in kt.items() if v[1]}}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", default="125000")
    ap.add_argument("--horizon", type=int, default=50)
    ap.add_argument("--storage", default="f32")
    ap.add_argument("--lib", default="probe", help="lib/libmppi_hip_<lib>.so (a build with -DMPPI_PROBE_TIMELINE)")
    ap.add_argument("--options", default="", help="name=value,... ; several sets separated by '/'")
    a = ap.parse_args()
    sets = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in s.split(",") if kv) for s in (a.options.split("/") if a.options else [""])]
    for K in [int(x) for x in a.samples.split(",")]:
        for opts in sets:
            print(json.dumps(run(K, a.horizon, a.storage, opts)), flush=True)
