#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into one small text summary."""
import csv, glob, os, sys, collections
root = sys.argv[1]
out = []
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    out.append("== kernel stats (%s)" % os.path.relpath(f, root))
    for row in csv.DictReader(open(f)):
        if "zwz" in row.get("Name", ""):
            out.append("%-60s calls=%s total_ns=%s avg_ns=%s pct=%s" % (row["Name"][:60], row.get("Calls"), row.get("TotalDurationNs"), row.get("AverageNs"), row.get("Percentage")))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob(os.path.join(root, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "zwz" not in k: continue
        k = k.split("(")[0][:40]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] in ("SQ_WAVES", "SQ_WAIT_INST_ANY", "FETCH_SIZE", "WRITE_SIZE"): cnt[(k, row["Counter_Name"])] += 1
out.append("== PMC sums over all dispatches (per kernel)")
for k, d in agg.items():
    out.append(k)
    for c, v in sorted(d.items()):
        out.append("    %-24s %.6g" % (c, v))
# HBM traffic per launch (MI355X_MICROARCH.md, HBM: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 on gfx950,
# FETCH_SIZE counts 64 B per 128 B request; counters from separate --pmc passes)
import json
calls = {}
for f in glob.glob(os.path.join(root, "pmc3", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "zwz" in k and row["Counter_Name"] == "FETCH_SIZE":
            calls[k.split("(")[0][:40]] = calls.get(k.split("(")[0][:40], 0) + 1
traffic = {}
for k, d in agg.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d and calls.get(k):
        traffic[k.replace("zwz::", "").replace("_kernel", "")] = {
            "launches": calls[k], "fetch_bytes_per_launch": int(2 * d["FETCH_SIZE"] * 1024 / calls[k]),
            "write_bytes_per_launch": int(d["WRITE_SIZE"] * 1024 / calls[k]),
            "hbm_bytes_per_launch": int((2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024 / calls[k])}
json.dump(traffic, open(os.path.join(root, "traffic.json"), "w"), indent=1)
out.append("== HBM traffic per launch (2*FETCH_SIZE + WRITE_SIZE)")
for k, v in traffic.items():
    out.append("%-12s %s" % (k, v))
open(os.path.join(root, "summary.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
