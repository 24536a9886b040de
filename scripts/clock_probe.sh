#!/bin/bash
# Diagnostics: sample the GPU clocks / power while the encode kernel runs.
python scripts/occupancy_sweep.py 6144 6144 6144 > gpurun_out/clk_sweep.log 2>&1 &
PID=$!
sleep 20
for i in 1 2 3 4 5 6; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk" | head -6
  echo ---
  sleep 1.5
done
wait $PID
tail -3 gpurun_out/clk_sweep.log
