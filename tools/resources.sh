#!/bin/bash
# Registers, scratch, LDS and occupancy of every kernel (hipcc cross-compiles gfx950 on the CPU box): part of every kernel change --
# a spill in one kernel has cost another kernel of the same file milliseconds before.   usage: tools/resources.sh [file.hip ...]
cd "$(dirname "$0")/../parallel-data-compression-and-decompression_amd"
for f in ${@:-csrc/zwz_kernels.hip csrc/zwz_band.hip csrc/zwz_plan.hip}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $EXTRA -Rpass-analysis=kernel-resource-usage -c $f -o /dev/null 2>&1 | python3 -c '
import re, sys
cur = {}
for line in sys.stdin:
    m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r":\d+:\d+: remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
    elif ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
        if k.strip().startswith("LDS Size"):
            n = re.sub(r"^_ZN3zwz\d+", "", cur["name"]); n = re.sub(r"E(Pv|PK|Pj|v|I).*$", "", n)
            print("%-34s VGPR %-4s AGPR %-3s SGPR %-4s scratch %-5s occupancy %-2s LDS %s" % (n[:34], cur.get("VGPRs"), cur.get("AGPRs"), cur.get("SGPRs"), cur.get("ScratchSize [bytes/lane]"), cur.get("Occupancy [waves/SIMD]"), cur.get("LDS Size [bytes/block]")))
'
done
