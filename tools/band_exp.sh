#!/bin/bash
# GPU-box helper: time the text workload's kernels for experimental builds of zwz_band.hip (-DZWZ_BAND_EXP=<bits>: results are
# wrong, only the durations mean anything).  usage: tools/band_exp.sh "<bits> <bits> ..." [files]
R=$GRAFT_REPO_ROOT; F=${2:-2000}
cd $R/parallel-data-compression-and-decompression_amd
for X in $1; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 -DZWZ_BAND_EXP=$X -shared -o libzwz_hip.so csrc/zwz_kernels.hip csrc/zwz_band.hip csrc/zwz_plan.hip csrc/zwz_api.cpp csrc/zwz_host.cpp csrc/zwz_pipeline.cpp 2> /dev/null || { echo build failed; exit 1; }
  (cd /tmp && export TMPDIR=/tmp && O=$R/gpurun_out/bandexp_$X && mkdir -p $O && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --workload text --files $F > $O/trace.log 2>&1
   python3 - <<PY
import csv, glob, collections
dur = collections.defaultdict(list)
for f in glob.glob("$O/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "band" in r["Kernel_Name"] or "sort" in r["Kernel_Name"]: dur[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("EXP $X", {k.replace("zwz::", ""): round(max(v) / 1e6, 3) for k, v in dur.items()})
PY
  )
done
