#!/usr/bin/env python3
"""Same-box A/B of BUILDS of the library (make -C motion_planning_amd/csrc VARIANT=name EXTRA="-D..." -> lib/libmppi_hip_<name>.so):
one process per (round, library), alternating, each running tools/ab_option.py's protocol on the given workload.

    python tools/ab_lib.py --libs default,nt,sc1 [--samples K] [--horizon T] [--agents A] [--co-shards 1] [--storage f32] [--rounds 2]
"default" = the product library.  One JSON line per run."""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", required=True)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--child", default="")
    a, rest = ap.parse_known_args()
    if a.child:
        sys.path.insert(0, ROOT)
        from motion_planning_amd import _capi
        if a.child != "default":
            _capi.LIB_PATH = os.path.join(ROOT, "motion_planning_amd", "lib", "libmppi_hip_%s.so" % a.child)
        import importlib.util
        spec = importlib.util.spec_from_file_location("ab_option", os.path.join(ROOT, "tools", "ab_option.py"))
        ab = importlib.util.module_from_spec(spec); spec.loader.exec_module(ab)
        p = argparse.ArgumentParser()
        for name, typ, dflt in (("--samples", int, 1000000), ("--horizon", int, 50), ("--agents", int, 1), ("--storage", str, "f32"), ("--tick-path", str, "auto"),
                                ("--co-shards", int, 1), ("--samples-total", int, 0), ("--ticks", int, 300), ("--fixed", str, "")):
            p.add_argument(name, type=typ, default=dflt)
        b = p.parse_args(rest)
        b.fixed_options = dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in b.fixed.split(",") if kv)
        r = ab.run("lanes_zero_copy", 1, b)     # (an option at its default value: the protocol needs one)
        r.update({"lib": a.child, "option": None, "value": None})
        print(json.dumps(r), flush=True)
    else:
        for rnd in range(a.rounds):
            for lib in a.libs.split(","):
                out = subprocess.run([sys.executable, os.path.abspath(__file__), "--libs", a.libs, "--child", lib] + rest, capture_output=True, text=True)
                line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else json.dumps({"lib": lib, "error": out.stderr[-400:]})
                print(json.dumps(dict(json.loads(line), round=rnd)), flush=True)
