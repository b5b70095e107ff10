#!/bin/bash
# usage (through gpurun, repo root): bash tools/final_r03b.sh  -- bench line, predicted shard makespans (all-chain rule on / off for 8 shards)
python -m scrna_parameter_estimation_amd.build > /dev/null 2>&1 || { echo build failed; exit 1; }
timeout -k 10 500 python bench.py > gpurun_out/r03b_bench_line.json 2> gpurun_out/r03b_bench.err; tail -c 2500 gpurun_out/r03b_bench_line.json
for n in 2 4 8; do
  timeout -k 10 300 python bench.py --predict-shards $n --steps 2 --warmup 1 > gpurun_out/r03b_predict_shards_$n.json 2>> gpurun_out/r03b_bench.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r03b_predict_shards_$n.json"))
print($n, d["predicted_ms_per_step"], d["predicted_value"], [(s["ms_per_step"], s["chain_waves"], s["tile_waves"]) for s in d["shards"]])
PY
done
MM_CHAIN_ALL_MAX=0 timeout -k 10 300 python bench.py --predict-shards 8 --steps 2 --warmup 1 > gpurun_out/r03b_predict_shards_8_tiles.json 2>> gpurun_out/r03b_bench.err
python - <<PY
import json
d = json.load(open("gpurun_out/r03b_predict_shards_8_tiles.json"))
print("8 (no all-chain rule)", d["predicted_ms_per_step"], d["predicted_value"], [(s["ms_per_step"], s["chain_waves"], s["tile_waves"]) for s in d["shards"]])
PY
