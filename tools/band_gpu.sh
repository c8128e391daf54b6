#!/bin/bash
# GPU-box helper while working on lz_sort / lz_match_band: the match-flavour parity tests, then the text workload under rocprofv3 (kernel times).
# usage: tools/band_gpu.sh <tag> [files]
TAG=${1:-b}; F=${2:-2000}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R && timeout -k 10 500 python -m pytest tests/test_gpu_codec.py -m gpu -x -q -k "flavours or matches_oracle or corners or golden" > gpurun_out/band_$TAG.log 2>&1; rc=$?
tail -3 gpurun_out/band_$TAG.log
[ $rc -ne 0 ] && { echo "GPU TESTS FAILED rc=$rc"; tail -40 gpurun_out/band_$TAG.log; exit $rc; }
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/bandprof_$TAG; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --workload text --files $F > $O/trace.log 2>&1 || { echo "bench failed"; tail -20 $O/trace.log; exit 1; }
python3 - <<PY
import csv, glob, collections
dur = collections.defaultdict(list)
for f in glob.glob("$O/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "zwz" in r["Kernel_Name"]: dur[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(dur.items(), key=lambda kv: -max(kv[1])):
    print("%-36s calls=%d max_ms=%.3f" % (k, len(v), max(v) / 1e6))
PY
tail -c 400 $O/trace.log
