#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into one small text summary."""
import csv, glob, os, sys, collections
root = sys.argv[1]
out = []
# A context's self-tests launch the product's own kernels on a handful of chunks (lz_links: 3 workgroups): they are left
# out of every per-launch figure -- only dispatches of the batch's grid size (the largest seen per kernel) count.
def short(name):
    """zwz::lz_sort_kernel(...) / void zwz::inflate_kernel<false>(...) -> zwz::lz_sort_kernel / zwz::inflate_kernel (template arguments dropped:
    the serial-header flavour of inflate only runs in a context's self-test and is filtered out by its grid size)"""
    name = name.split("(")[0]
    if name.startswith("void "): name = name[5:]
    return name.split("<")[0][:40]
def batch_rows(rows, name_key, grid_key):
    rows = [r for r in rows if "zwz" in r.get(name_key, "")]
    top = collections.defaultdict(int)
    for r in rows:
        top[short(r[name_key])] = max(top[short(r[name_key])], int(r[grid_key]))
    return [r for r in rows if int(r[grid_key]) * 4 >= top[short(r[name_key])]]
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
    out.append("== kernel stats over the batch launches (%s)" % os.path.relpath(f, root))
    dur = collections.defaultdict(list)
    for r in batch_rows(list(csv.DictReader(open(f))), "Kernel_Name", "Grid_Size_X"):
        dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = sum(sum(v) for v in dur.values()) or 1
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        out.append("%-60s calls=%d total_ns=%d avg_ns=%.1f pct=%.2f" % (k[:60], len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / tot))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob(os.path.join(root, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for row in batch_rows(list(csv.DictReader(open(f))), "Kernel_Name", "Grid_Size"):
        k = row.get("Kernel_Name", "")
        k = short(k)
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
    for row in batch_rows(list(csv.DictReader(open(f))), "Kernel_Name", "Grid_Size"):
        k = row.get("Kernel_Name", "")
        if row["Counter_Name"] == "FETCH_SIZE":
            calls[short(k)] = calls.get(short(k), 0) + 1
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
