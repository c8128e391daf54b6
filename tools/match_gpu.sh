#!/bin/bash
# GPU-box helper while working on lz_match: codec parity tests (production build), the incompressible workload's stage times, then the
# instrumented build's phase shares (tools/match_times.sh rebuilds the library: run last).   usage: tools/match_gpu.sh <tag> [files]
TAG=${1:-m}; F=${2:-10000}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R && timeout -k 10 700 python -m pytest tests/test_gpu_codec.py -m gpu -x -q > gpurun_out/match_$TAG.log 2>&1; rc=$?
tail -2 gpurun_out/match_$TAG.log
[ $rc -ne 0 ] && { echo "GPU TESTS FAILED rc=$rc"; tail -60 gpurun_out/match_$TAG.log; exit $rc; }
for w in "random --files $F" "small_files --files 30000"; do
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $w > gpurun_out/match_${TAG}.json 2> gpurun_out/match_${TAG}.err || { echo "bench failed"; tail -20 gpurun_out/match_${TAG}.err; exit 1; }
python3 - "$w" gpurun_out/match_${TAG}.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], "value=%s ok=%s compress=%.2f GB/s" % (d["value"], d["verified"]["ok"], d["compress_GBps"]), d["stage_ms_per_pass"])
PY
done
bash $R/tools/match_times.sh 4000 | grep MATCH_TIMES | head -1
