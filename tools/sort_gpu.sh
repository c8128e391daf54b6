#!/bin/bash
# GPU-box helper while working on lz_sort: codec parity tests (the context's self-test compares lz_sort's dest[] with a host counting sort),
# then kernel times of the text and small-file workloads.
TAG=${1:-s}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R && timeout -k 10 700 python -m pytest tests/test_gpu_codec.py -m gpu -x -q > gpurun_out/sort_$TAG.log 2>&1; rc=$?
tail -2 gpurun_out/sort_$TAG.log
[ $rc -ne 0 ] && { echo "GPU TESTS FAILED rc=$rc"; tail -60 gpurun_out/sort_$TAG.log; exit $rc; }
cd /tmp && export TMPDIR=/tmp
for w in "text --files 4000" "small_files --files 30000"; do
O=$R/gpurun_out/trace_sort_$TAG; rm -rf $O; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --workload $w > $O/trace.log 2>&1 || { echo "trace failed"; tail -20 $O/trace.log; exit 1; }
echo "== $w"
python3 - <<PY
import csv, glob, collections
dur = collections.defaultdict(list)
for f in glob.glob("$O/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "zwz" in r["Kernel_Name"]: dur[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(dur.items(), key=lambda kv: -max(kv[1])):
    if "sort" in k or "place" in k or "band" in k: print("%-36s calls=%d max_ms=%.3f" % (k, len(v), max(v) / 1e6))
PY
done
