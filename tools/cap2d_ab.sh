#!/bin/bash
# usage (through gpurun, repo root): bash tools/cap2d_ab.sh  -- the 2D replay (configs[3] share: 250 x 2000 pairs, 500k cells, 1,000 bootstraps) with / without the attempt cap
for f in "-DBOOT_BTPE_CAP=0" "-DBOOT_BTPE_CAP=1"; do
  MM_EXTRA_DEFS="$f" python -m scrna_parameter_estimation_amd.build --force > /dev/null 2>&1 || { echo build failed; exit 1; }
  echo "== flags: $f"
  timeout -k 10 400 python tools/bench_2d.py 500000 8000 250 2000 1000 2>&1 | grep "pairs=\|pair kernels\|packing"
done
MM_EXTRA_DEFS="" python -m scrna_parameter_estimation_amd.build --force > /dev/null 2>&1
timeout -k 10 300 python -m pytest tests/test_gpu_api.py tests/test_gpu_configs.py -q -x -k "2d or c4 or corr" 2>&1 | tail -3
