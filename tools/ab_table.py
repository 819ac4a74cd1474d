#!/usr/bin/env python3
"""Prints the jsonl files of tools/fused_ab.py as a table."""
import json, sys
for f in sys.argv[1:]:
    for l in open(f):
        d = json.loads(l); k = d["kernels_us"]
        print("%-10s co%d %-14s tick %7.1f us  ran %s ok %s ds %.1e du %.1e V %s | %s" % (
            d["config"], d["co_shards"], d["variant"], d["tick_us"], d["fused_ran"], d["ok"], d["max_abs_diff_state"], d["max_abs_diff_u"],
            d["V_bit_identical"], " ".join("%s %.1f" % (a, b) for a, b in k.items() if b)))
