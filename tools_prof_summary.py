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
open(os.path.join(root, "summary.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
