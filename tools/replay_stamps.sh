#!/bin/bash
# usage (through gpurun, repo root): bash tools/replay_stamps.sh [config]
MM_EXTRA_DEFS="-DBOOT_STAMPS" python -m scrna_parameter_estimation_amd.build --force > /dev/null 2>&1 || { echo build failed; exit 1; }
python tools/replay_stamps.py ${1:-C3} 2>&1 | tail -14
MM_EXTRA_DEFS="" python -m scrna_parameter_estimation_amd.build --force > /dev/null 2>&1
