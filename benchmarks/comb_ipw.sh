#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in 13 14; do
  BPPP_RP_COMB_BITS=$c BPPP_RP_NO_SPLIT=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pc -o t -- python benchmarks/prove_timing.py 4096 > gpurun_out/pc.log 2>&1
  python - "$c" <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open('gpurun_out/pc/t_kernel_trace.csv')) if 'k_comb_msm' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6 for r in rows]
b=[r for r in csv.DictReader(open('gpurun_out/pc/t_kernel_trace.csv')) if 'k_comb_mult' in r['Kernel_Name']]
print("c", sys.argv[1], "comb_msm ms (last batch):", [round(x,2) for x in d[-11:]], "sum", round(sum(d[-11:]),1), "build", [(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6 for r in b])
PY
  grep "total ms" gpurun_out/pc.log | tail -1
done
