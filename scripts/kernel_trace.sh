# kernel-trace only: bash scripts/kernel_trace.sh <tag> <bench args>
cd /tmp && export TMPDIR=/tmp
R=/root/repo
TAG=$1; shift
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python3 $R/bench.py "$@" > $O/trace.log 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob("$O/trace/**/t_kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print("$TAG", r["Name"][:60], r["Calls"], "%.1f ms avg" % (float(r["AverageNs"])/1e6), r["Percentage"])
PY
