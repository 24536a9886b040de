"""Sum rocprofv3 --pmc counters per kernel (counter_collection.csv) -- python scripts/pmc_by_kernel.py DIR [DIR...]"""
import collections, csv, glob, sys
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[1 if r["Kernel_Name"].startswith("(") else 0]
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].split("<")[0]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
        for k, v in sorted(agg.items()):
            if max(v.values()) < 1e6: continue
            print(f, k, "calls", len(calls[k]), {a: "%.4g" % b for a, b in sorted(v.items())})
