#!/usr/bin/env python3
"""How much of a co-scheduled tick's kernels overlap in time, from a rocprofv3 --kernel-trace CSV (CPU: no GPU needed).

    python tools/trace_overlap.py gpurun_out/final/stats_c5/*/*_kernel_trace.csv > profiles/rN_trace_overlap_c5.json
Takes the steady-state half of the trace, keeps the engine's kernels (mppi::...), and reports per kernel name and grid size (the shards of a co-scheduled handle and the one-engine leg of the same command differ in it): launches, mean
duration, and the share of its running time during which a kernel of ANOTHER queue (the other co-scheduled engine) was running
too; plus the wall-clock share of the window with 0 / 1 / >= 2 engine kernels in flight."""
import csv, json, sys
from collections import defaultdict


def main(path):
    rows = [r for r in csv.DictReader(open(path)) if "mppi::" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[len(rows) // 2:]                                     # steady state (the timed legs come last)
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0] + " [grid %s x %s]" % (r["Grid_Size_X"], r["Grid_Size_Y"])) for r in rows]
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    # sweep line over start / end events
    pts = []
    for i, (s, e, q, n) in enumerate(ev):
        pts.append((s, 1, i)); pts.append((e, -1, i))
    pts.sort()
    active, last = set(), t0
    depth_ns = defaultdict(int)
    other_ns = defaultdict(int)          # per kernel index: ns during which another queue's kernel ran
    for t, d, i in pts:
        dt = t - last
        if dt > 0:
            depth_ns[min(len(active), 2)] += dt
            qs = defaultdict(int)
            for j in active:
                qs[ev[j][2]] += 1
            for j in active:
                if len(qs) > 1 or qs[ev[j][2]] < len(active):
                    other_ns[j] += dt
        last = t
        if d == 1: active.add(i)
        else: active.discard(i)
    per = defaultdict(lambda: [0, 0, 0])
    for i, (s, e, q, n) in enumerate(ev):
        per[n][0] += 1; per[n][1] += e - s; per[n][2] += other_ns[i]
    tot = float(t1 - t0)
    out = {"trace": path.split("/")[-1], "window_us": tot / 1e3, "queues": sorted({e[2] for e in ev}),
           "wall_share_by_kernels_in_flight": {("0", "1", ">=2")[k]: v / tot for k, v in sorted(depth_ns.items())},
           "per_kernel": {n: {"launches": c, "mean_us": d / c / 1e3, "share_of_its_time_next_to_another_queues_kernel": o / d if d else 0.0}
                          for n, (c, d, o) in per.items()}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
