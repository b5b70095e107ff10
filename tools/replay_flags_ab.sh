#!/bin/bash
# usage (through gpurun, repo root): bash tools/replay_flags_ab.sh "<defs A>" "<defs B>" ...   -- C3 bootstrap time for each set of -D flags
for D in "$@"; do
  MM_EXTRA_DEFS="$D" python -m scrna_parameter_estimation_amd.build --force > /dev/null 2>&1 || { echo "build failed: $D"; exit 1; }
  echo "== $D"
  timeout -k 10 300 python tools/pack_sweep.py C3 "220,3,2000" 2>&1 | tail -1 | cut -c1-120
done
MM_EXTRA_DEFS="" python -m scrna_parameter_estimation_amd.build --force > /dev/null 2>&1
