# bash scripts/bench_level.sh <level> <blocks> [ENV=VAL ...]
L=$1; B=$2; shift 2
for kv in "$@"; do export "$kv"; done
python bench.py --level $L --blocks $B --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/lz_tmp.json 2>gpurun_out/lz_tmp.err
python - <<PY
import json
try:
    j=json.loads(open("gpurun_out/lz_tmp.json").read().strip().splitlines()[-1]); print("level $L blocks $B $*", "value", j["value"], "ms", j["ms_per_step"], "enc GB/s", j["roofline"]["achieved"])
except Exception as e: print("level $L $*", "ERR", e)
PY
