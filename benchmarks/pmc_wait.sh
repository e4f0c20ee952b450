#!/bin/bash
# wait-state counters of the two madd-bound kernels: k_comb_msm (prover) and k_acc_points (2^20 MSM)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BPPP_RP_NO_SPLIT=1 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pcm -o t -- python3 benchmarks/prove_timing.py 4096 > gpurun_out/pcm.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pcm2 -o t -- python3 bench.py --headline-only --no-cpu-baseline > gpurun_out/pcm2.log 2>&1
python3 - <<'PY'
import csv, collections, glob
for d in ('pcm','pcm2'):
    f=glob.glob('gpurun_out/%s/*counter_collection.csv'%d)[0]
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'k_comb_msm' in k or 'k_acc_points' in k:
            acc[k+' grid='+r['Grid_Size']][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in list(acc.items())[:4]:
        m={c: sum(x)/len(x) for c,x in v.items()}
        print(k, {c: round(x/1e6,1) for c,x in m.items()}, "wait_inst/wave_cycles %.2f valu/wave_cycles %.3f" % (m['SQ_WAIT_INST_ANY']/m['SQ_WAVE_CYCLES'], m['SQ_INSTS_VALU']/m['SQ_WAVE_CYCLES']))
PY
