#!/bin/bash
# usage (through gpurun, repo root): bash tools/rec2d_ab.sh  -- the 2D replay on per-chain records (one BTPE attempt per bin step) against shared rows; tests first
python -m scrna_parameter_estimation_amd.build > /dev/null 2>&1 || { echo build failed; exit 1; }
timeout -k 10 400 python -m pytest tests/test_gpu_api.py tests/test_gpu_configs.py tests/test_gpu_kernels.py -q -x -k "2d or c4 or corr or pair" 2>&1 | tail -3
for v in 1 0; do
  echo "== MM_BOOT2D_RECORDS=$v"
  MM_BOOT2D_RECORDS=$v timeout -k 10 400 python tools/bench_2d.py 500000 8000 250 2000 1000 2>&1 | grep "pairs=\|pair kernels"
done
