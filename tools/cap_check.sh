#!/bin/bash
# usage (through gpurun, repo root): bash tools/cap_check.sh  -- bit-identity tests of the replay kernels, then the launch at C3 / C2 with the default packing
python -m scrna_parameter_estimation_amd.build > /dev/null 2>&1 || { echo build failed; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_properties.py -x -q 2>&1 | tail -3
timeout -k 10 300 python tools/chain_sweep.py C3 lone 2>&1 | grep -A12 "^setting"
timeout -k 10 300 python tools/chain_sweep.py C2 lone 2>&1 | grep -A12 "^setting"
