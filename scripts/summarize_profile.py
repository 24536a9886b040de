"""Turn gpurun_out/prof_<tag>/ (scripts/gpu_profile.sh) into the committed summaries under profiles/."""
import collections, csv, glob, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 6144
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
note = sys.argv[3] if len(sys.argv) > 3 else None
level = int(sys.argv[4]) if len(sys.argv) > 4 else 1
out = os.path.join(root, "profiles"); os.makedirs(out, exist_ok=True)

import re
def short(k):
    """The library's kernels by their base name (template arguments and the namespace dropped): k_encode_rec, k_decode_rec,
    k_hc12_search, k_encode_rec_hc, k_encode_rec_dict, k_decode_rec_linked, ..."""
    m = re.search(r"\b(k_[a-z0-9_]+)", k)
    return m.group(1) if m else None

stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
summary = {"tag": tag, "blocks_per_gpu": B, "level": level, "kernels": {}}
if note: summary["workload"] = note
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    keep = [r for r in rows if short(r["Name"])]
    with open(os.path.join(out, "%s_kernel_stats.csv" % tag), "w") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep)
    for r in keep:
        summary["kernels"][short(r["Name"])] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                                                "min_ms": float(r["MinNs"]) / 1e6, "max_ms": float(r["MaxNs"]) / 1e6}
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("fetch", "write", "sq1", "sq2"):
    for f in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k: pmc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
# one rocprofv3 row per dispatch and counter (already summed over XCDs/SEs)
for k, cs in pmc.items():
    e = summary["kernels"].setdefault(k, {})
    e["pmc_per_launch"] = {c: sum(v) / len(v) for c, v in cs.items()}
    e["pmc_launches"] = {c: len(v) for c, v in cs.items()}
    e["pmc_sum"] = {c: sum(v) for c, v in cs.items()}
# A bench step is one ABI call per direction, but an HC call launches its kernels once per GROUP of blocks: per-step totals =
# the sum over all launches of a pass / the steps of that pass.  k_scan (the frame body's scan) runs exactly once per step.
steps_by_counter = summary["kernels"].get("k_scan", {}).get("pmc_launches", {})
summary["steps_in_trace"] = summary["kernels"].get("k_scan", {}).get("calls")
S = B * (4 << 20)
for k in list(summary["kernels"]):
    p = summary["kernels"].get(k, {}).get("pmc_per_launch", {})
    if "FETCH_SIZE" in p and "WRITE_SIZE" in p:
        # rocprofv3 reports KiB.  MI355X_MICROARCH.md §HBM: FETCH_SIZE reads 1/2 of the bytes of wide coalesced streams on
        # gfx950 (exact for 16 B/lane streaming); our access mix (16 B/lane unaligned windows + random candidate reads + byte
        # stores) is uncalibrated, so both the raw and the x2-corrected figure are kept.
        raw = (p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024
        cor = (2 * p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024
        # what the counters see is traffic across the L2 boundary: Infinity Cache hits included, so an upper bound of HBM bytes
        summary["kernels"][k]["l2_miss_traffic_bytes_raw"] = raw
        summary["kernels"][k]["l2_miss_traffic_bytes"] = cor
        summary["kernels"][k]["plaintext_bytes"] = S
        nf, nw = steps_by_counter.get("FETCH_SIZE"), steps_by_counter.get("WRITE_SIZE")
        ps = summary["kernels"][k]["pmc_sum"]
        if nf and nw:
            summary["kernels"][k]["l2_miss_traffic_bytes_per_step"] = (2 * ps["FETCH_SIZE"] / nf + ps["WRITE_SIZE"] / nw) * 1024
            summary["kernels"][k]["launches_per_step"] = summary["kernels"][k]["pmc_launches"]["FETCH_SIZE"] / nf
for k in summary["kernels"]:
    summary["kernels"][k].pop("pmc_sum", None)
json.dump(summary, open(os.path.join(out, "%s_summary.json" % tag), "w"), indent=1)
print(json.dumps(summary, indent=1))
