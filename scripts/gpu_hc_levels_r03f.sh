# Bench lines of HC levels 3..11 at 4096 blocks with the builder beside the walk (levels 3, 9 with the CPU baseline)
# -> gpurun_out/r03f_bench_level*.json
set -x
for l in 3 9; do
  timeout -k 10 420 python bench.py --level $l --blocks 4096 --steps 1 --warmup 1 > gpurun_out/r03f_bench_level${l}_B4096.json 2> gpurun_out/r03f_bench_level${l}.err || exit 1
done
for l in 4 5 6 7 8 10 11; do
  timeout -k 10 300 python bench.py --level $l --blocks 4096 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r03f_bench_level${l}_B4096.json 2> gpurun_out/r03f_bench_level${l}.err || exit 1
done
python - <<'PY'
import json
for l in (3,4,5,6,7,8,9,10,11):
    j=json.loads(open("gpurun_out/r03f_bench_level%d_B4096.json"%l).read().strip().splitlines()[-1])
    print(l, j["value"], j["ms_per_step"], j["ms"]["encode_kernel"], (j.get("cpu_baseline") or {}).get("value"))
PY
