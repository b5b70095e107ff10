#!/bin/bash
# usage (through gpurun, repo root): bash tools/cap_pack.sh  -- the capped-attempt tile kernel with the re-swept packing: per-width timings at C3, constants at C2
python -m scrna_parameter_estimation_amd.build > /dev/null 2>&1 || { echo build failed; exit 1; }
timeout -k 10 300 python tools/chain_sweep.py C3 lone 2>&1 | grep -A12 "^setting"
timeout -k 10 400 python tools/pack_sweep.py C2 220,3,2000 260,2.2,2000 240,2.6,2000 280,1.9,2000 260,2.2,2000,2048,1024,3050 220,3,2000 260,2.2,2000 2>&1 | grep "C0="
