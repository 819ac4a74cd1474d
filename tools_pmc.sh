#!/bin/bash
# usage (on the GPU box): tools_pmc.sh <outdir-tag> [bench args...]
# one rocprofv3 --pmc pass per counter group (never combined with sys/hip tracing)
R=$GRAFT_REPO_ROOT; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/pmc$i -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/$TAG/pmc$i.log 2>&1
done
python3 - <<PY
import csv,glob,collections,os
R=os.environ['GRAFT_REPO_ROOT']
for d in sorted(glob.glob(R+'/gpurun_out/$TAG/pmc*/*/*counter_collection.csv')):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(d)):
        k=r['Kernel_Name'].split('(')[0][:40]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k in acc:
        if 'rollout' in k or 'update' in k:
            print(k, {c: (sum(v)/len(v)) for c,v in acc[k].items()})
PY
