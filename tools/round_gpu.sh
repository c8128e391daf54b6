#!/bin/bash
# GPU-box helper: the whole -m gpu suite, the 40 000-chunk parity soak, the 370 000-file kernel-scope bench and a kernel trace
# of the 30 000-file one.   usage: tools/round_gpu.sh <tag> [soak-seed]
TAG=${1:-x}; SEED=${2:-3}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R && timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/suite_$TAG.log 2>&1; rc=$?
tail -3 gpurun_out/suite_$TAG.log
[ $rc -ne 0 ] && { echo "GPU TESTS FAILED rc=$rc"; tail -60 gpurun_out/suite_$TAG.log; exit $rc; }
timeout -k 10 600 python tools/soak_gpu.py 40000 $SEED > gpurun_out/soak_$TAG.log 2>&1 || { echo "soak failed"; tail -20 gpurun_out/soak_$TAG.log; exit 1; }
tail -2 gpurun_out/soak_$TAG.log
timeout -k 10 600 python bench.py --workload small_files --no-cpu-baseline > gpurun_out/small_files_370000_$TAG.json 2> gpurun_out/small_files_370000_$TAG.err || { echo "small_files failed"; tail -20 gpurun_out/small_files_370000_$TAG.err; exit 1; }
python3 - gpurun_out/small_files_370000_$TAG.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("small_files 370000: value=%s ok=%s compress=%s decompress=%s" % (d["value"], d["verified"]["ok"], d["compress_GBps"], d["decompress_GBps"]), d["stage_ms_per_pass"])
PY
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/trace_small_$TAG; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --workload small_files --files 30000 > $O/trace.log 2>&1 || { echo "trace failed"; tail -20 $O/trace.log; exit 1; }
python3 - <<PY
import csv, glob, collections
dur = collections.defaultdict(list)
for f in glob.glob("$O/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "zwz" in r["Kernel_Name"]: dur[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(dur.items(), key=lambda kv: -max(kv[1])):
    print("%-36s calls=%d max_ms=%.3f" % (k, len(v), max(v) / 1e6))
PY
