#!/bin/bash
# usage (through gpurun, repo root): bash tools/cap5_ab.sh  -- configs[4] (ht_1d_vs_control, many-chain regime) with / without the attempt cap
for f in "-DBOOT_BTPE_CAP=0" "-DBOOT_BTPE_CAP=1"; do
  MM_EXTRA_DEFS="$f" python -m scrna_parameter_estimation_amd.build --force > /dev/null 2>&1 || { echo build failed; exit 1; }
  echo "== flags: $f"
  timeout -k 10 500 python tools/bench_vs_control.py 2>&1 | tail -4
done
MM_EXTRA_DEFS="" python -m scrna_parameter_estimation_amd.build --force > /dev/null 2>&1
