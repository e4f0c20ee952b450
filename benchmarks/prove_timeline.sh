#!/bin/bash
# kernel timeline of the LAST bppp_rp_prove_batch call of benchmarks/prove_timing.py per batch size (rocprofv3 kernel trace)
# usage: BPPP_RP_COMB_MIN=1 benchmarks/prove_timeline.sh "1 64" [marker kernel]   -> gpurun_out/pt_<B>.txt
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
for B in ${1:-"1"}; do
  rm -rf gpurun_out/pt_$B
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pt_$B -o t -- python3 benchmarks/prove_timing.py $B > gpurun_out/pt_$B.log 2>&1 || { tail -n 5 gpurun_out/pt_$B.log; exit 1; }
  f=$(find gpurun_out/pt_$B -name "*kernel_trace.csv" | head -n 1)
  python3 benchmarks/timeline.py "$f" ${2:-k_rpp_draws} 0 > gpurun_out/pt_$B.txt
  grep "total ms" gpurun_out/pt_$B.log | tail -n 2 >> gpurun_out/pt_$B.txt
  rm -rf gpurun_out/pt_$B
done
