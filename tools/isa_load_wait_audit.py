#!/usr/bin/env python3
"""CPU: the order of vector-memory loads (L), stores (S), barriers (B) and vmcnt waits (Wn) in every kernel of a translation unit, from the
compiler's own assembly -- one line per kernel.  A run like  LL(W0)LL(W0)LL(W0)...  is a kernel waiting for each pair of loads before it
issues the next (loads behind per-lane guards each get their own basic block and their own s_waitcnt): what held the update kernel at
5.4 TB/s until round 5 (EXPERIMENTS.md 50).

    python tools/isa_load_wait_audit.py [motion_planning_amd/csrc/mppi_engine.hip ...] [-D...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def audit(src, defs=()):
    """{kernel name (demangled, without arguments): {"lines", "vgprs", "scratch", "seq"}} for one translation unit."""
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                    src, "-o", out] + list(defs), check=True, stderr=subprocess.DEVNULL)
    L = open(out).read().splitlines()
    os.unlink(out)
    res = {}
    for s, l in enumerate(L):
        if not (l.startswith("_ZN4mppi") and ":" in l and not l.startswith("\t")):
            continue
        e = next(i for i in range(s + 1, len(L)) if L[i].startswith(".Lfunc_end"))
        seq = []
        for x in L[s:e]:
            t = x.strip()
            if t.startswith(("global_load", "buffer_load", "flat_load")): seq.append("L")
            elif t.startswith("s_waitcnt") and "vmcnt" in t: seq.append("(W%s)" % re.search(r"vmcnt\((\d+)\)", t).group(1))
            elif t.startswith("s_barrier"): seq.append("B")
            elif t.startswith(("global_store", "buffer_store", "flat_store")): seq.append("S")
        tail = L[e:e + 60]
        num = lambda key: next((int(y.split(":")[1]) for y in tail if y.strip().startswith("; " + key + ":")), None)
        name = subprocess.run(["c++filt", l.split(":")[0]], capture_output=True, text=True).stdout.strip().split("(")[0]
        res[name.replace("void ", "")] = {"lines": e - s, "vgprs": num("NumVgprs"), "scratch": num("ScratchSize"), "seq": "".join(seq),
                                          "spills": sum("scratch_" in x for x in L[s:e])}
    return res


if __name__ == "__main__":
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]
    srcs = [a for a in sys.argv[1:] if not a.startswith("-D")] or [os.path.join(ROOT, "motion_planning_amd", "csrc", "mppi_engine.hip")]
    for s in srcs:
        for name, r in audit(s, defs).items():
            s2 = re.sub(r"(B)\1{3,}", lambda m: "B*%d" % len(m.group(0)), r["seq"])
            print("%-70s %5d lines  %3s VGPRs  scratch %s  %s" % (name[:70], r["lines"], r["vgprs"], r["scratch"], s2[:400]))
