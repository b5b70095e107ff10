#!/bin/bash
# usage (through gpurun, repo root): bash tools/free_check.sh  -- the free-running tile kernel (MM_TILE_FREE=1): bit-identity tests, then the launch at C3 / C2 beside the default
python -m scrna_parameter_estimation_amd.build > /dev/null 2>&1 || { echo build failed; exit 1; }
MM_TILE_FREE=1 timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_properties.py -x -q 2>&1 | tail -3
for v in 1 0; do
  echo "== MM_TILE_FREE=$v"
  MM_TILE_FREE=$v timeout -k 10 300 python tools/chain_sweep.py C3 lone 2>&1 | grep -A12 "^setting"
done
MM_TILE_FREE=1 timeout -k 10 300 python tools/chain_sweep.py C2 lone 2>&1 | grep "^setting"
