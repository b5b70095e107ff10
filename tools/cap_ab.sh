#!/bin/bash
# usage (through gpurun, repo root): bash tools/cap_ab.sh "<flags>" ... -- in-kernel stamps of the tile kernel per build flag set
for f in "$@"; do
  MM_EXTRA_DEFS="-DBOOT_STAMPS $f" python -m scrna_parameter_estimation_amd.build --force > /dev/null 2>&1 || { echo build failed; exit 1; }
  echo "== $f"
  timeout -k 10 300 python tools/replay_stamps.py C3 2>&1 | tail -9
done
MM_EXTRA_DEFS="" python -m scrna_parameter_estimation_amd.build --force > /dev/null 2>&1
