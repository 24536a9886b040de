# Round 4, final state: the bench lines of every HC level at 4096 blocks again (the decoder behind them is this round's), levels 2, 3, 9, 12
# with the CPU baseline.  -> gpurun_out/r04b_bench_level*_B4096.json
for l in 2 3 9 12; do
  timeout -k 10 500 python bench.py --level $l --blocks 4096 --steps 1 --warmup 1 > gpurun_out/r04b_bench_level${l}_B4096.json 2> gpurun_out/r04b_bench_level${l}.err || exit 1
done
for l in 4 5 6 7 8 10 11; do
  timeout -k 10 300 python bench.py --level $l --blocks 4096 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r04b_bench_level${l}_B4096.json 2> gpurun_out/r04b_bench_level${l}.err || exit 1
done
python - <<'PY'
import json
for l in range(2, 13):
    j=json.loads(open("gpurun_out/r04b_bench_level%d_B4096.json"%l).read().strip().splitlines()[-1])
    print(l, j["value"], j["ms_per_step"], j["ms"]["encode_kernel"], j["ms"]["decode_kernel"], (j.get("cpu_baseline") or {}).get("value"), j["roofline"].get("traffic_source"))
PY
