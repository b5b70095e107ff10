#!/bin/bash
# usage (through gpurun, repo root): bash tools/flags_ab.sh "<flags A>" "<flags B>" ...   -- build-flag A/B of the replay bootstrap at C3, same box
for f in "$@"; do
  MM_EXTRA_DEFS="$f" python -m scrna_parameter_estimation_amd.build --force > /dev/null 2>&1 || { echo build failed; exit 1; }
  echo "== flags: $f"
  timeout -k 10 300 python tools/chain_sweep.py C3 lone 2>&1 | grep "setting"
done
MM_EXTRA_DEFS="" python -m scrna_parameter_estimation_amd.build --force > /dev/null 2>&1
