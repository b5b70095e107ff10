#!/bin/bash
# usage (through gpurun, repo root): bash tools/clock_watch.sh  -- shader clock / power while the C3 replay launch runs (sampled through rocm-smi)
python tools/chain_sweep.py C3 lone lone lone > gpurun_out/clock_watch_run.log 2>&1 &
PID=$!
for i in $(seq 1 60); do
  sleep 1
  if ! kill -0 $PID 2>/dev/null; then break; fi
  /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power\|mclk" | tr '\n' ' '
  echo
done
wait $PID
grep "^setting" gpurun_out/clock_watch_run.log
