#!/bin/bash
# Every rocprofv3 pass whose summary is committed under profiles/ (then: python benchmarks/summarise_profiles.py r03).
# Counter passes carry --pmc only (no trace domains); the program itself follows `--`.
#   usage: benchmarks/profile_round.sh [stats] [msmpmc] [provepmc]     (default: all three groups)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
G=gpurun_out
what=${*:-"stats msmpmc provepmc"}
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
  esac
done
# keep only the small summaries (the merge back is capped at 64 MiB): *_kernel_stats.csv and per-kernel reductions of the counter files
python3 benchmarks/summarise_profiles.py --reduce
find $G -name "*_kernel_trace.csv" -delete; find $G -name "*counter_collection.csv" -delete; find $G -name "*agent_info.csv" -delete
