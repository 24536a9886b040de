#!/bin/bash
# GPU check of the HC levels: parity tests, then a short bench line per strategy (mid / hash chain / optimal).
set -u
mkdir -p gpurun_out
python -m pytest tests/test_hc.py -m gpu -x -q > gpurun_out/hc_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/hc_tests.log
tail -3 gpurun_out/hc_tests.log
for lvl in 2 3 6 9; do
  timeout 900 python bench.py --level $lvl --blocks 1024 --steps 1 --warmup 1 > gpurun_out/hc_bench_l$lvl.json 2> gpurun_out/hc_bench_l$lvl.err
  echo "level $lvl rc=$?"; tail -c 1500 gpurun_out/hc_bench_l$lvl.json
done
