# Round 4: PMC profiles of levels 2, 3 and 9 at 4096 blocks (per-step traffic: an HC call launches its kernels once per group), then
# the bench lines of every HC level at 4096 blocks (levels 2, 3, 9, 12 with the CPU baseline), whose traffic fields name the summary
# of THEIR level or stay null.  -> gpurun_out/prof_r04_level*_B4096/, gpurun_out/r04_bench_level*_B4096.json
set -x
for l in 3 9 2; do
  bash scripts/gpu_profile_r04.sh r04_level${l}_B4096 full --level $l --blocks 4096 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/p_l$l.txt 2>&1
  python scripts/summarize_profile.py r04_level${l}_B4096 4096 "level $l, 4096 x 4MiB T blocks, one step (bench.py --level $l --steps 1 --warmup 0)" $l > /dev/null
done
for l in 2 3 9 12; do
  timeout -k 10 500 python bench.py --level $l --blocks 4096 --steps 1 --warmup 1 > gpurun_out/r04_bench_level${l}_B4096.json 2> gpurun_out/r04_bench_level${l}.err || exit 1
done
for l in 4 5 6 7 8 10 11; do
  timeout -k 10 300 python bench.py --level $l --blocks 4096 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r04_bench_level${l}_B4096.json 2> gpurun_out/r04_bench_level${l}.err || exit 1
done
python - <<'PY'
import json
for l in range(2, 13):
    j=json.loads(open("gpurun_out/r04_bench_level%d_B4096.json"%l).read().strip().splitlines()[-1])
    print(l, j["value"], j["ms_per_step"], j["ms"]["encode_kernel"], (j.get("cpu_baseline") or {}).get("value"), j["roofline"].get("traffic"), j["roofline"].get("traffic_source"))
PY
