#!/usr/bin/env python3
"""Static VALU instruction mix of the rollout kernel's steady-state loop, from the compiler's own assembly.

    make -C motion_planning_amd/csrc asm          # writes build/asm/rollout_f32_n4-...-gfx950.s
    python tools/valu_mix.py [--symbol SUBSTR] [--steps-per-iter 6] > profiles/rN_valu_mix.json

Finds the kernel whose mangled name contains SUBSTR, takes its largest self-looping basic block (the full
6-step chunk of the lane-per-sample rollout) and counts wave-instructions per issue class.  The per-class issue
costs are the ones tools/ubench.hip measured on MI355X (cycles per wave64 instruction per SIMD, 4 waves/SIMD):
bench.py turns count x cost into the VALU-issue roofline of the kernel."""
import argparse
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# cycles per wave64 instruction per SIMD at the MEASURED shader clock (tools/ubench.hip reads s_memtime against
# s_memrealtime around every microbenchmark; profiles/r3_ubench.txt, 4 waves per SIMD).  Round 2's table was the same
# measurements scaled by an ASSUMED 1.9 GHz and read ~10 % low (fp64 3.9 instead of 4.35).
COST = {"f64": 4.35, "trans_f32": 8.2, "mad_u64_u32": 4.33, "pk_f32": 4.25, "cvt_f64": 4.2, "half_rate_f32": 4.25, "other_valu": 2.45}


def classify(op):
    if not op.startswith("v_"):
        return None
    if op.startswith("v_pk_"):
        return "pk_f32"
    if op.startswith(("v_cvt_f64", "v_cvt_f32_f64")):
        return "cvt_f64"
    if op.endswith("_f64") or "_f64_" in op:
        return "f64"
    if re.match(r"v_(exp|log|sqrt|sin|cos|rcp|rsq)_f32", op):
        return "trans_f32"
    if op.startswith(("v_mad_u64_u32", "v_mad_i64_i32")):
        return "mad_u64_u32"
    # measured at ~4.2 cycles although single-precision: v_med3 / v_max / v_min, everything with a DPP operand, permlane swaps
    if re.match(r"v_(med3|min3|max3|max|min)_f32", op) or op.endswith("_dpp") or op.startswith("v_permlane"):
        return "half_rate_f32"
    return "other_valu"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm", default=os.path.join(ROOT, "build", "asm", "rollout_f32_n4-hip-amdgcn-amd-amdhsa-gfx950.s"))
    # float storage, NTERM 4, Philox, eps not stored, inline nominal (one wave, T <= 64), rk4 model, the node's cost
    ap.add_argument("--symbol", default="rollout_kernelIfLi4ELb1ELb0ELi1ELi0ELb0E")
    # sample-steps per loop iteration: 6 for rollout_kernel, 12 for rollout_pk_kernel (six steps of two samples per lane)
    ap.add_argument("--steps-per-iter", type=int, default=6)
    args = ap.parse_args()
    lines = open(args.asm).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and args.symbol in l.split(":")[0])
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
    # loops: a backward branch to a label defined earlier in the function; body = everything in between
    # (a chunk of the rollout loop spans several compiler basic blocks).  The largest one is the full-chunk loop.
    label_at, insts = {}, []
    for l in lines[start:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            label_at[m.group(1)] = len(insts)
            continue
        t = l.strip()
        if t and not t.startswith((";", ".", "//")):
            insts.append(t)
    loops = []
    for i, ins in enumerate(insts):
        op = ins.split()[0]
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = ins.split()[-1]
            if tgt in label_at and label_at[tgt] <= i:
                loops.append((label_at[tgt], i, tgt))
    # innermost loops only (no other loop nested inside); the largest of them is the straight-line full chunk
    inner = [l for l in loops if not any(o is not l and o[0] >= l[0] and o[1] <= l[1] for o in loops)]
    if not inner:
        sys.exit("no loop found")
    # the steady-state loop of FULL blocks is the one whose stores are raw buffer stores (the ragged-block variant of
    # the same chunk guards plain global stores with exec masks); ties -> the larger body
    def key(l):
        ops = [i.split()[0] for i in insts[l[0]:l[1] + 1]]
        return (sum(o.startswith("buffer_store") for o in ops), l[1] - l[0])
    # rollout_pk_kernel: the full-block loop stores one 8-byte pair per lane and step, its ragged-block twin two dwords
    paired = [l for l in inner if any(i.split()[0] == "buffer_store_dwordx2" for i in insts[l[0]:l[1] + 1])]
    best = max(paired or inner, key=key)
    n, body = best[2], insts[best[0]:best[1] + 1]
    counts = {}
    n_salu = n_vmem = n_lds = n_other = 0
    for ins in body:
        op = ins.split()[0]
        c = classify(op)
        if c:
            counts[c] = counts.get(c, 0) + 1
        elif op.startswith("s_"):
            n_salu += 1
        elif op.startswith(("buffer_", "global_", "flat_")):
            n_vmem += 1
        elif op.startswith("ds_"):
            n_lds += 1
        else:
            n_other += 1
    valu = sum(counts.values())
    cyc = sum(COST[k] * v for k, v in counts.items())
    out = {"symbol": args.symbol, "loop_block": n, "steps_per_iteration": args.steps_per_iter,
           "valu_per_iteration": valu, "valu_per_step": valu / args.steps_per_iter,
           "by_class_per_iteration": counts, "cost_cycles_per_class": COST,
           "issue_cycles_per_iteration": cyc, "issue_cycles_per_step": cyc / args.steps_per_iter,
           "avg_cycles_per_valu": cyc / valu, "salu": n_salu, "vmem": n_vmem, "lds": n_lds, "other": n_other,
           "source": "hipcc -save-temps assembly of " + os.path.basename(args.asm) + "; costs from tools/ubench.hip"}
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
