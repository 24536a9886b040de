# Round-3 bench lines of the HC levels at 4096 blocks (levels 2, 3, 9 with the CPU baseline) -> gpurun_out/r03_bench_level*.json
set -x
for l in 2 3 9; do
  timeout -k 10 420 python bench.py --level $l --blocks 4096 --steps 1 --warmup 1 > gpurun_out/r03_bench_level${l}_B4096.json 2> gpurun_out/r03_bench_level${l}.err || exit 1
done
for l in 4 5 6 7 8 10 11; do
  timeout -k 10 300 python bench.py --level $l --blocks 4096 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_level${l}_B4096.json 2> gpurun_out/r03_bench_level${l}.err || exit 1
done
timeout -k 10 400 python bench.py --level 12 --blocks 4096 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r03_bench_config4_level12_B4096.json 2> gpurun_out/r03_bench_level12.err || exit 1
python - <<'PY'
import json
for l in (2,3,4,5,6,7,8,9,10,11):
    j=json.loads(open("gpurun_out/r03_bench_level%d_B4096.json"%l).read().strip().splitlines()[-1])
    print(l, j["value"], j["ms_per_step"], j["roofline"]["frac"], (j.get("cpu_baseline") or {}).get("value"))
PY
