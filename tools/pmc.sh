#!/bin/bash
# Run ON THE GPU BOX (via gpurun): tools/pmc.sh <tag> [bench args...]
# One rocprofv3 --pmc pass per counter group (never combined with sys/hip tracing), then a
# summary JSON (per-launch averages per kernel) next to the raw CSVs under gpurun_out/<tag>/.
R=$GRAFT_REPO_ROOT; TAG=$1; shift
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/pmc$i -- python $R/bench.py --steps 8 --warmup 2 --min-warmup-s 0 --no-cpu-baseline --no-f64-line "$@" > $R/gpurun_out/$TAG/pmc$i.log 2>&1
done
python3 - <<PY
import csv,glob,collections,os,json
R=os.environ['GRAFT_REPO_ROOT']
out=collections.defaultdict(dict)
for d in sorted(glob.glob(R+'/gpurun_out/$TAG/pmc*/*/*counter_collection.csv')):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(d)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k in acc:
        for c,v in acc[k].items():
            out[k][c]=sum(v)/len(v)
            out[k]['launches_'+c]=len(v)
summ={}
for k,c in out.items():
    if "mppi::" not in k: continue
    e=dict(c)
    if 'FETCH_SIZE' in c: e['hbm_read_bytes_per_launch (FETCH_SIZE KB x 1024 x 2: gfx950 wide-load correction, MI355X_MICROARCH.md)']=c['FETCH_SIZE']*1024*2
    if 'WRITE_SIZE' in c: e['hbm_write_bytes_per_launch (WRITE_SIZE KB x 1024; calibrated on the 12 B/step stored-eps rollout = 604 MB)']=c['WRITE_SIZE']*1024
    summ[k]=e
json.dump(summ, open(R+'/gpurun_out/$TAG/pmc_summary.json','w'), indent=1, sort_keys=True)
for k in summ:
    if 'rollout' in k or 'update' in k: print(k[:50], {a:round(b) for a,b in summ[k].items() if 'bytes' in a or a in ('SQ_INSTS_VALU','SQ_WAVES','SQ_ACTIVE_INST_VALU','GRBM_GUI_ACTIVE','TCC_HIT_sum','TCC_MISS_sum')})
PY
