# timeline of the two-pipeline experiment: bash scripts/exp_pingpong_trace.sh BLOCKS
cd /tmp && export TMPDIR=/tmp
R=/root/repo
B=$1
O=$R/gpurun_out/pp_trace_$B
mkdir -p $O
EXP_SHARED_OUT=1 timeout -k 10 400 rocprofv3 --kernel-trace -d $O/trace -o t --output-format csv -- python3 $R/scripts/exp_pingpong.py $B 6 2 > $O/trace.log 2>&1
python3 - <<PY
import csv,glob
rows=[]
for f in glob.glob("$O/trace/**/t_kernel_trace.csv", recursive=True):
    rows+=list(csv.DictReader(open(f)))
rows=[r for r in rows if any(k in r["Kernel_Name"] for k in ("k_l1_","k_scan_from","k_decode"))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[0]["Start_Timestamp"])
out=open("$O/timeline.txt","w")
for r in rows[-60:]:
    line="%-28s q%-3s start %9.2f ms  end %9.2f ms  dur %8.2f" % (r["Kernel_Name"].split("(")[0][-28:], r.get("Queue_Id","?"), (int(r["Start_Timestamp"])-t0)/1e6, (int(r["End_Timestamp"])-t0)/1e6, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
    print(line); out.write(line+"\n")
PY
