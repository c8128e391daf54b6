#!/bin/bash
# GPU-box helper: kernel durations (rocprofv3 --kernel-trace) of one bench workload.  usage: tools/trace_gpu.sh <workload> <files> <tag>
W=${1:-random}; F=${2:-2000}; TAG=${3:-t}
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/trace_${TAG}_$W; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload $W --files $F > $O/trace.log 2>&1 || { echo "bench failed"; tail -20 $O/trace.log; exit 1; }
python3 - <<PY
import csv, glob, collections
dur = collections.defaultdict(list)
for f in glob.glob("$O/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "zwz" in r["Kernel_Name"]: dur[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(dur.items(), key=lambda kv: -max(kv[1])):
    v2 = sorted(v)
    print("%-36s calls=%d max_ms=%.3f median_ms=%.3f" % (k, len(v), max(v) / 1e6, v2[len(v2)//2] / 1e6))
PY
