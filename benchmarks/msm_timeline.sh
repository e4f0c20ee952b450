#!/bin/bash
# kernel timeline of ONE bppp_msm_device call per size (rocprofv3 kernel trace; the program itself after `--`)
# usage: benchmarks/msm_timeline.sh "22016 65536 344838"   -> gpurun_out/mt_<n>.txt
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
for N in ${1:-"65536"}; do
  rm -rf gpurun_out/mt_$N
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/mt_$N -o t -- python3 benchmarks/msm_timing.py $N > gpurun_out/mt_$N.log 2>&1 || { tail -5 gpurun_out/mt_$N.log; exit 1; }
  f=$(find gpurun_out/mt_$N -name "*kernel_trace.csv" | head -1)
  python3 benchmarks/timeline.py "$f" ${2:-k_digits} 0 > gpurun_out/mt_$N.txt
  grep "msm ms" gpurun_out/mt_$N.log | tail -n 2 >> gpurun_out/mt_$N.txt
  rm -rf gpurun_out/mt_$N
done
