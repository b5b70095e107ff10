#!/bin/bash
# usage (through gpurun, repo root): bash tools/final_r03.sh  -- the round's closing measurements: bench line with --extra, shard predictions
python -m scrna_parameter_estimation_amd.build > /dev/null 2>&1 || { echo build failed; exit 1; }
timeout -k 10 500 python bench.py --steps 3 --warmup 1 --extra > gpurun_out/r03f_bench_line.json 2> gpurun_out/r03f_bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r03f_bench_line.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["bootstrap"]["ms"], d["bootstrap"]["lane_occupancy"], d.get("value_e2e"), d.get("pair_tests_per_s"), d.get("vs_control_tests_per_s"), d["roofline"]["frac"], d["cpu_baseline"]["value"], d["cpu_baseline"]["max_rel_p_diff"])
PY
for n in 2 4 8; do
  timeout -k 10 300 python bench.py --predict-shards $n --steps 2 --warmup 1 > gpurun_out/r03f_predict_shards_$n.json 2>> gpurun_out/r03f_bench.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r03f_predict_shards_$n.json"))
print($n, d["predicted_ms_per_step"], d["predicted_value"], [s["ms_per_step"] for s in d["shards"]])
PY
done
