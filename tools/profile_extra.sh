#!/bin/bash
# Kernel summaries of BASELINE configs[3] (one GPU's share of the 2000 x 2000 pair grid) and configs[4] (Perturb-seq) on the GPU
# box (run through gpurun from the repo root):  bash tools/profile_extra.sh <tag>
# Copy what should be judged from gpurun_out/ into profiles/ afterwards.
set -e -o pipefail
TAG=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_c4prof -- python3 $R/tools/bench_2d.py 500000 8000 250 2000 1000 > $OUT/${TAG}_c4prof.log 2>&1
tail -3 $OUT/${TAG}_c4prof.log
python3 $R/tools/stats_summary.py $OUT/${TAG}_c4prof > $OUT/${TAG}_2d_kernels.txt; cat $OUT/${TAG}_2d_kernels.txt
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_c5prof -- python3 $R/tools/bench_vs_control.py 200000 15000 500 5000 1 > $OUT/${TAG}_c5prof.log 2>&1
tail -3 $OUT/${TAG}_c5prof.log
python3 $R/tools/stats_summary.py $OUT/${TAG}_c5prof > $OUT/${TAG}_vs_control_kernels.txt; cat $OUT/${TAG}_vs_control_kernels.txt
