#!/bin/bash
# usage (through gpurun, repo root): bash tools/chain_prio_ab.sh "prio list" settings...   -- chain-kernel wave priority A/B at C3
PRIOS=$1; shift
for p in $PRIOS; do
  MM_EXTRA_DEFS="-DCHAIN_PRIO=$p" python -m scrna_parameter_estimation_amd.build --force > /dev/null 2>&1 || { echo build failed; exit 1; }
  echo "== CHAIN_PRIO $p"
  timeout -k 10 500 python tools/chain_sweep.py C3 "$@" 2>&1 | grep -v amdgpu.ids | grep "setting\|span"
done
MM_EXTRA_DEFS="" python -m scrna_parameter_estimation_amd.build --force > /dev/null 2>&1
