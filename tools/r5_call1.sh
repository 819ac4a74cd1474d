#!/bin/bash
# Round 5, GPU call 1 (run ON the GPU box): the whole GPU suite on the new build, then the first diagnostics and same-box A/Bs.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c1; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests -m gpu -x -q 2>&1 ) > $O/gpu_tests.log 2>&1; tail -5 $O/gpu_tests.log
timeout 60 ./tools/depbench > $O/depbench.txt 2>&1; tail -40 $O/depbench.txt
AB="timeout 120 python tools/ab_option.py --rounds 2 --ticks 300"
OPTS="low_occ=0,table_hoist=0/low_occ=1,table_hoist=0/low_occ=0,table_hoist=1/low_occ=1,table_hoist=1"
timeout 200 python tools/probe_timeline.py --samples 125000,250000 --options $OPTS > $O/timeline_T50.jsonl 2>$O/timeline.err
timeout 200 python tools/probe_timeline.py --samples 100000 --horizon 100 --options $OPTS > $O/timeline_T100.jsonl 2>>$O/timeline.err
timeout 100 python tools/probe_timeline.py --samples 500000,1000000 --options table_hoist=0/table_hoist=1 > $O/timeline_pk.jsonl 2>>$O/timeline.err
for K in 125000 250000; do
  $AB --option table_hoist --values 0,1 --samples $K --fixed low_occ=0 > $O/ab_hoist_$K.jsonl 2>>$O/ab.err
  $AB --option low_occ --values 0,1 --samples $K --fixed table_hoist=0 > $O/ab_lowocc_$K.jsonl 2>>$O/ab.err
  $AB --option low_occ --values 0,1 --samples $K --fixed table_hoist=1 > $O/ab_lowocc_hoisted_$K.jsonl 2>>$O/ab.err
done
$AB --option table_hoist --values 0,1 --samples 100000 --horizon 100 --fixed low_occ=0 > $O/ab_hoist_c3.jsonl 2>>$O/ab.err
$AB --option low_occ --values 0,1 --samples 100000 --horizon 100 --fixed table_hoist=1 > $O/ab_lowocc_hoisted_c3.jsonl 2>>$O/ab.err
$AB --option table_hoist --values 0,1 --samples 500000 > $O/ab_hoist_500000.jsonl 2>>$O/ab.err
$AB --option table_hoist --values 0,1 --samples 1000000 > $O/ab_hoist_c4_one_engine.jsonl 2>>$O/ab.err
$AB --option table_hoist --values 0,1 --samples 1000000 --co-shards 0 > $O/ab_hoist_c4_co.jsonl 2>>$O/ab.err
$AB --option table_hoist --values 0,1 --samples 16384 --agents 64 > $O/ab_hoist_c5_one_engine.jsonl 2>>$O/ab.err
$AB --option fin_threads --values 0,256,512 --samples 16384 --agents 64 > $O/ab_fin_c5_one_engine.jsonl 2>>$O/ab.err
$AB --option fin_threads --values 0,256,512 --samples 16384 --agents 64 --co-shards 0 > $O/ab_fin_c5_split.jsonl 2>>$O/ab.err
$AB --option fin_threads --values 0,256,512 --samples 125000 > $O/ab_fin_125000.jsonl 2>>$O/ab.err
$AB --option fin_threads --values 0,256,512 --samples 10000 > $O/ab_fin_c2.jsonl 2>>$O/ab.err
timeout 200 python tools/ab_option.py --rounds 2 --ticks 200 --option k_pieces --values 1,2,3,4 --samples 1000000 --storage f64 > $O/ab_pieces_f64.jsonl 2>>$O/ab.err
timeout 200 python tools/ab_option.py --rounds 2 --ticks 200 --option k_pieces --values 1,2,3 --samples 1000000 --storage f32 --fixed rollout_pk=0 > $O/ab_pieces_f32_fp64kernel.jsonl 2>>$O/ab.err
timeout 200 python bench.py --no-cpu-baseline --steps 100 2>$O/bench.err | tail -1 > $O/bench_c4.json
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/ab_*.jsonl")):
    rows=[json.loads(l) for l in open(f) if l.strip()]
    by={}
    for r in rows: by.setdefault(r["value"],[]).append(r)
    print(os.path.basename(f), {v:[round(x["tick_us"],1) for x in rs] for v,rs in by.items()}, {v:{k:round(x,1) for k,x in rs[-1]["bracketed_us"].items()} for v,rs in by.items()})
for f in sorted(glob.glob("$O/timeline_*.jsonl")):
    for l in open(f):
        r=json.loads(l); print(os.path.basename(f), r["K"], r["T"], r["options"], r["kernel"], "clk", round(r["clock_mhz"]), "total", r["wave_total_cycles"], "deltas", r["deltas_cycles"], r["bracketed_us"])
PY
