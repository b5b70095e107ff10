#!/bin/bash
# Timing-only ablations of the replay kernel (wrong results by construction): rebuilds the library with -D flags on the GPU box.
# usage (through gpurun, repo root): bash tools/replay_ablate.sh [config]
CFG=${1:-C3}
for defs in "" "-DNPY_ABLATE_BTPE" "-DNPY_ABLATE_INV_LOOP" "-DNPY_ABLATE_BTPE -DNPY_ABLATE_INV_LOOP"; do
  MM_EXTRA_DEFS="$defs" python -m scrna_parameter_estimation_amd.build --force > /dev/null 2>&1 || { echo "build failed for $defs"; exit 1; }
  echo "== defs: '$defs'"
  python tools/pack_sweep.py $CFG 220,3,2000 2>&1 | grep "C0="
done
MM_EXTRA_DEFS="" python -m scrna_parameter_estimation_amd.build --force > /dev/null 2>&1
