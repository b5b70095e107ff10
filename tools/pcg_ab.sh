#!/bin/bash
# usage (through gpurun, repo root): bash tools/pcg_ab.sh  -- issue costs of the instructions involved, then the replay launch with
# and without the generator's 128-bit multiply (timing only)
hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_cost tools/valu_cost.hip 2>/dev/null && timeout -k 10 120 /tmp/valu_cost
bash tools/flags_ab.sh "" "-DNPY_ABLATE_PCG"
