#!/bin/bash
# usage (through gpurun, repo root): bash tools/free_ab.sh "<flags>" ...  -- the record-based tile kernel (MM_TILE_FREE=1) per build flag set, C3
for f in "$@"; do
  MM_EXTRA_DEFS="$f" python -m scrna_parameter_estimation_amd.build --force > /dev/null 2>&1 || { echo build failed; exit 1; }
  echo "== flags: $f"
  MM_TILE_FREE=1 timeout -k 10 300 python tools/chain_sweep.py C3 lone 2>&1 | grep "^setting\|tiles of\|free-running"
done
MM_EXTRA_DEFS="" python -m scrna_parameter_estimation_amd.build --force > /dev/null 2>&1
