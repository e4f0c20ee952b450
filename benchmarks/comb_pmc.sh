#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BPPP_RP_NO_SPLIT=1 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pcm -o t -- python3 benchmarks/prove_timing.py 4096 > gpurun_out/pcm.log 2>&1
python3 - <<'PY'
import csv, collections, glob
f=glob.glob('gpurun_out/pcm/*counter_collection.csv')[0]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0]
    if 'k_comb_msm' in k or 'k_acc_points' in k:
        acc[k+' grid='+r['Grid_Size']][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    print(k, {c: round(sum(x)/len(x)) for c,x in v.items()}, 'n', len(next(iter(v.values()))))
PY
