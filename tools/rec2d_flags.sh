#!/bin/bash
# usage (through gpurun, repo root): bash tools/rec2d_flags.sh "<flags>" ...  -- the record-based 2D replay per build flag set (configs[3] share)
for f in "$@"; do
  MM_EXTRA_DEFS="$f" python -m scrna_parameter_estimation_amd.build --force > /dev/null 2>&1 || { echo build failed; exit 1; }
  echo "== flags: $f"
  timeout -k 10 400 python tools/bench_2d.py 500000 8000 250 2000 1000 2>&1 | grep "pairs="
done
MM_EXTRA_DEFS="" python -m scrna_parameter_estimation_amd.build --force > /dev/null 2>&1
