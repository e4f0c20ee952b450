#!/bin/bash
# kernel timeline of ONE bppp_rp_verify_batch_device call per batch size (rocprofv3 kernel trace; the program itself after `--`)
# usage: benchmarks/verify_timeline.sh "1 256 4096"   -> gpurun_out/vt_<B>.txt
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
for B in ${1:-"1 256 4096"}; do
  rm -rf gpurun_out/vt_$B
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/vt_$B -o t -- python3 benchmarks/verify_timing.py $B > gpurun_out/vt_$B.log 2>&1 || { tail -5 gpurun_out/vt_$B.log; exit 1; }
  f=$(find gpurun_out/vt_$B -name "*kernel_trace.csv" | head -1)
  python3 benchmarks/timeline.py "$f" k_rp_decode_points 0 > gpurun_out/vt_$B.txt
  grep "verify ms" gpurun_out/vt_$B.log | tail -2 >> gpurun_out/vt_$B.txt
  rm -rf gpurun_out/vt_$B
done
