#!/bin/bash
# usage (through gpurun, repo root): bash tools/ingest_ab.sh "<defs A>" "<defs B>" ...   -- K0 timing for each set of -D flags
for D in "$@"; do
  MM_EXTRA_DEFS="$D" python -m scrna_parameter_estimation_amd.build --force > /dev/null 2>&1 || { echo "build failed: $D"; exit 1; }
  echo "== $D"
  timeout -k 10 200 python tools/ingest_bench.py 2>&1 | tail -1
done
MM_EXTRA_DEFS="" python -m scrna_parameter_estimation_amd.build --force > /dev/null 2>&1
