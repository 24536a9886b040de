# TA / TCP / TCC counters of the level-1 kernels (serial steps, one stream), one small pass each: bash scripts/gpu_pmc_ta.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/prof_$1
mkdir -p $O
i=0
for set in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_BUSY_avr TCC_TAG_STALL_sum TCC_CYCLE_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o s --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 --duplex 0 --pipelines 1 > $O/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv,glob,collections,re
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m=re.search(r"\b(k_[a-z0-9_]+)", r["Kernel_Name"])
        if m: acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in ("k_l1_parse","k_decode_rec","k_l1_sizes","k_l1_write"):
    print(k, {c: "%.3e" % (sum(v)/len(v)) for c,v in sorted(acc[k].items())})
PY
