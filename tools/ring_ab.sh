#!/bin/bash
# usage (through gpurun, repo root): bash tools/ring_ab.sh "rounds list"   -- uniforms produced ahead per bin step in the tile kernel's ring mode, C3
for r in $1; do
  MM_EXTRA_DEFS="-DBOOT_RING_ROUNDS=$r" python -m scrna_parameter_estimation_amd.build --force > /dev/null 2>&1 || { echo build failed; exit 1; }
  echo "== BOOT_RING_ROUNDS $r"
  timeout -k 10 300 python tools/chain_sweep.py C3 lone 2>&1 | grep "setting"
done
MM_EXTRA_DEFS="" python -m scrna_parameter_estimation_amd.build --force > /dev/null 2>&1
