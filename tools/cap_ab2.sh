#!/bin/bash
# usage (through gpurun, repo root): bash tools/cap_ab2.sh "<flags>" ...  -- production build A/B of the replay launch at C3 with per-width tile timings
for f in "$@"; do
  MM_EXTRA_DEFS="$f" python -m scrna_parameter_estimation_amd.build --force > /dev/null 2>&1 || { echo build failed; exit 1; }
  echo "== flags: $f"
  timeout -k 10 300 python tools/chain_sweep.py C3 lone 2>&1 | grep -A12 "^setting"
done
MM_EXTRA_DEFS="" python -m scrna_parameter_estimation_amd.build --force > /dev/null 2>&1
