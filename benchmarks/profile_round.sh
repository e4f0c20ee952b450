#!/bin/bash
# Every rocprofv3 pass whose summary is committed under profiles/ (then: python benchmarks/summarise_profiles.py r03).
# Counter passes carry --pmc only (no trace domains); the program itself follows `--`.
#   usage: benchmarks/profile_round.sh [stats] [msmpmc] [provepmc] [verifypmc]     (default: all four groups)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
G=gpurun_out
what=${*:-"stats msmpmc provepmc verifypmc"}
run() { name=$1; shift; rm -rf $G/$name; "$@" > $G/$name.log 2>&1 || { echo "FAILED: $name"; tail -5 $G/$name.log; exit 1; }; echo "done: $name"; }
for w in $what; do
  case $w in
    stats)
      run prof_msm   rocprofv3 --kernel-trace --stats --output-format csv -d $G/prof_msm   -o msm   -- python3 bench.py --headline-only --no-cpu-baseline
      run prof_bench rocprofv3 --kernel-trace --stats --output-format csv -d $G/prof_bench -o bench -- python3 bench.py --no-cpu-baseline
      BPPP_RP_NO_SPLIT=1 run prof_prove rocprofv3 --kernel-trace --stats --output-format csv -d $G/prof_prove -o prove -- python3 benchmarks/prove_timing.py 4096
      ;;
    msmpmc)
      run pmc_fetch rocprofv3 --pmc FETCH_SIZE --output-format csv -d $G/pmc_fetch -o fetch -- python3 bench.py --headline-only --no-cpu-baseline
      run pmc_write rocprofv3 --pmc WRITE_SIZE --output-format csv -d $G/pmc_write -o write -- python3 bench.py --headline-only --no-cpu-baseline
      ;;
    provepmc)
      BPPP_RP_NO_SPLIT=1 run pmc_prove_fetch rocprofv3 --pmc FETCH_SIZE --output-format csv -d $G/pmc_prove_fetch -o pf -- python3 benchmarks/prove_timing.py 4096
      BPPP_RP_NO_SPLIT=1 run pmc_prove_write rocprofv3 --pmc WRITE_SIZE --output-format csv -d $G/pmc_prove_write -o pw -- python3 benchmarks/prove_timing.py 4096
      BPPP_RP_NO_SPLIT=1 run pmc_prove_sq rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU --output-format csv -d $G/pmc_prove_sq -o ps -- python3 benchmarks/prove_timing.py 4096
      ;;
    verifypmc)
      # the verifier alone: the proofs are made once (unprofiled) and read back by every pass
      SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"
      export VERIFY_REPS=6
      PROOF_FILES=$G/vt_4096.npz run vt_make python3 benchmarks/verify_timing.py 4096
      PROOF_FILES=$G/bin_1024.npz run bin_make python3 benchmarks/binary_64by64.py 1024
      PROOF_FILES=$G/vt_4096.npz run prof_verify rocprofv3 --kernel-trace --stats --output-format csv -d $G/prof_verify -o v -- python3 benchmarks/verify_timing.py 4096
      PROOF_FILES=$G/vt_4096.npz run pmc_verify_fetch rocprofv3 --pmc FETCH_SIZE --output-format csv -d $G/pmc_verify_fetch -o vf -- python3 benchmarks/verify_timing.py 4096
      PROOF_FILES=$G/vt_4096.npz run pmc_verify_write rocprofv3 --pmc WRITE_SIZE --output-format csv -d $G/pmc_verify_write -o vw -- python3 benchmarks/verify_timing.py 4096
      PROOF_FILES=$G/vt_4096.npz run pmc_verify_sq rocprofv3 --pmc $SQ --output-format csv -d $G/pmc_verify_sq -o vs -- python3 benchmarks/verify_timing.py 4096
      PROOF_FILES=$G/bin_1024.npz run prof_binv rocprofv3 --kernel-trace --stats --output-format csv -d $G/prof_binv -o b -- python3 benchmarks/binary_64by64.py 1024
      PROOF_FILES=$G/bin_1024.npz run pmc_binv_fetch rocprofv3 --pmc FETCH_SIZE --output-format csv -d $G/pmc_binv_fetch -o bf -- python3 benchmarks/binary_64by64.py 1024
      PROOF_FILES=$G/bin_1024.npz run pmc_binv_write rocprofv3 --pmc WRITE_SIZE --output-format csv -d $G/pmc_binv_write -o bw -- python3 benchmarks/binary_64by64.py 1024
      PROOF_FILES=$G/bin_1024.npz run pmc_binv_sq rocprofv3 --pmc $SQ --output-format csv -d $G/pmc_binv_sq -o bs -- python3 benchmarks/binary_64by64.py 1024
      # the new provers' kernel times (prove + verify in one trace)
      run prof_binp rocprofv3 --kernel-trace --stats --output-format csv -d $G/prof_binp -o bp -- python3 benchmarks/binary_64by64.py 1024
      run prof_ipp rocprofv3 --kernel-trace --stats --output-format csv -d $G/prof_ipp -o ip -- python3 benchmarks/ip_prove_timing.py 16384
      rm -f $G/vt_4096.npz $G/bin_1024.npz
      ;;
  esac
done
# keep only the small summaries (the merge back is capped at 64 MiB): *_kernel_stats.csv and per-kernel reductions of the counter files
python3 benchmarks/summarise_profiles.py --reduce
find $G -name "*_kernel_trace.csv" -delete; find $G -name "*counter_collection.csv" -delete; find $G -name "*agent_info.csv" -delete
