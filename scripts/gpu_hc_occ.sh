#!/bin/bash
# HC level 12 throughput against the number of resident waves per CU (encode only matters; bench prints enc MiB/s).
set -u
mkdir -p gpurun_out
for w in 8 16 32; do
  b=$((256 * w))
  PLZ4HIP_HC_WAVES_PER_CU=$w timeout 1500 python bench.py --level 12 --blocks $b --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/hc12_w$w.json 2> gpurun_out/hc12_w$w.err
  echo "waves/CU $w rc=$?"; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/hc12_w$w.json")); print("  blocks", d["config"]["blocks_per_gpu"], "enc MiB/s", d["enc_MiBps_per_gpu"], "value", d["value"], "ms", d["ms"])
except Exception as e:
    print("  no json:", e); print(open("gpurun_out/hc12_w$w.err").read()[-600:])
PY
done
