#!/bin/bash
# Per-round profiling on the GPU box (run through gpurun from the repo root):  bash tools/profile_round.sh <tag> [sq|traffic|stats ...]
#   sq      : SQ counters of every kernel of one C3 bench step (the replay bootstrap kernel is the one of interest)
#   traffic : FETCH_SIZE / WRITE_SIZE of K1 and the ingest kernels (two separate passes) -> gpurun_out/<tag>_k1_traffic_C3.json
#   stats   : rocprofv3 --kernel-trace --stats summary of the default bench command
# Copy what should be judged from gpurun_out/ into profiles/ afterwards.
set -e -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
for what in "$@"; do
  case $what in
    sq)
      timeout -k 10 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY \
        --output-format csv -d $OUT/${TAG}_pmc_sq -- python3 $R/bench.py --config C3 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/${TAG}_pmc_sq.log 2>&1
      python3 $R/tools/pmc_summary.py $OUT/${TAG}_pmc_sq > $OUT/${TAG}_sq_counters_C3.txt
      tail -3 $OUT/${TAG}_pmc_sq.log; cat $OUT/${TAG}_sq_counters_C3.txt ;;
    traffic)
      timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -- python3 $R/tools/k1_traffic.py run > $OUT/${TAG}_pmc_fetch.log 2>&1
      timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -- python3 $R/tools/k1_traffic.py run > $OUT/${TAG}_pmc_write.log 2>&1
      python3 $R/tools/k1_traffic.py parse $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_k1_traffic_C3.json ;;
    stats)
      timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $R/bench.py --config C3 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/${TAG}_stats.log 2>&1
      tail -2 $OUT/${TAG}_stats.log
      f=$(find $OUT/${TAG}_stats -name "*kernel_stats.csv" | head -1); cp $f $OUT/${TAG}_bench_C3_kernel_stats.csv; head -12 $f ;;
  esac
done
