#!/bin/bash
# Run ON THE GPU BOX (via gpurun): tools/pmc_fused.sh <tag>
# Where the time goes in the three forms of the tick (tools/fused_loop.py --fused 0 | 1 | 2): one rocprofv3 --pmc pass per counter
# group and form (never combined with sys/hip tracing), kernel-trace for the durations, then one summary JSON of per-launch averages.
R=$GRAFT_REPO_ROOT; TAG=$1; shift
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in 0 1 2; do
  i=0
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
             "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/m${mode}_pmc$i -- python $R/tools/fused_loop.py --fused $mode --ticks 12 "$@" > $O/m${mode}_pmc$i.log 2>&1
  done
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/m${mode}_stats -- python $R/tools/fused_loop.py --fused $mode --ticks 200 "$@" > $O/m${mode}_stats.log 2>&1
done
python3 - <<PY
import csv,glob,collections,os,json
O="$O"
summ={}
for mode in (0,1,2):
    out=collections.defaultdict(dict)
    for d in sorted(glob.glob(O+'/m%d_pmc*/*/*counter_collection.csv' % mode)):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(d)):
            k=r['Kernel_Name'].split('(')[0].replace('void ','')
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        for k in acc:
            for c,v in acc[k].items():
                out[k][c]=sum(v)/len(v); out[k]['launches']=len(v)
    for f in glob.glob(O+'/m%d_stats/*/*kernel_stats.csv' % mode):
        for r in csv.DictReader(open(f)):
            k=r['Name'].split('(')[0].replace('void ','')
            if k in out: out[k]['avg_us_kernel_trace']=float(r['AverageNs'])/1e3; out[k]['calls_kernel_trace']=int(r['Calls'])
    tick=None
    try: tick=json.loads(open(O+'/m%d_stats.log' % mode).read().strip().splitlines()[-1])
    except Exception: pass
    summ["fused=%d" % mode]={"tick_under_rocprof": tick, "kernels": {k:v for k,v in out.items() if 'mppi::' in k}}
json.dump(summ, open(O+'/pmc_fused_summary.json','w'), indent=1, sort_keys=True)
for m,v in summ.items():
    print(m, v["tick_under_rocprof"])
    for k,c in v["kernels"].items():
        if any(x in k for x in ("rollout","update","fused")):
            print("   ", k[:70], {a: round(b,1) for a,b in c.items() if a in ("avg_us_kernel_trace","SQ_WAVES","SQ_INSTS_VALU","SQ_ACTIVE_INST_VALU","SQ_BUSY_CYCLES","SQ_WAVE_CYCLES","SQ_WAIT_INST_ANY","GRBM_GUI_ACTIVE","FETCH_SIZE","WRITE_SIZE")})
PY
