#!/bin/bash
# usage (through gpurun, repo root): bash tools/ingest_stamps.sh
MM_EXTRA_DEFS="-DINGEST_STAMPS" python -m scrna_parameter_estimation_amd.build --force > /dev/null 2>&1 || { echo build failed; exit 1; }
python tools/ingest_stamps.py 2>&1 | tail -6
MM_EXTRA_DEFS="" python -m scrna_parameter_estimation_amd.build --force > /dev/null 2>&1
