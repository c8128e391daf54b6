#!/bin/bash
# GPU-box helper while working on inflate: codec parity tests, stage times of text / small files / random, then the instrumented build's phase shares.
TAG=${1:-i}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R && timeout -k 10 700 python -m pytest tests/test_gpu_codec.py -m gpu -x -q > gpurun_out/inflate_$TAG.log 2>&1; rc=$?
tail -2 gpurun_out/inflate_$TAG.log
[ $rc -ne 0 ] && { echo "GPU TESTS FAILED rc=$rc"; tail -60 gpurun_out/inflate_$TAG.log; exit $rc; }
for w in "text --files 4000" "small_files --files 30000" "random --files 4000"; do
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $w > gpurun_out/inflate_${TAG}.json 2> gpurun_out/inflate_${TAG}.err || { echo "bench failed"; tail -20 gpurun_out/inflate_${TAG}.err; exit 1; }
python3 - "$w" gpurun_out/inflate_${TAG}.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], "value=%s ok=%s decompress=%.2f GB/s inflate=%.3f ms" % (d["value"], d["verified"]["ok"], d["decompress_GBps"], d["stage_ms_per_pass"]["inflate"]))
PY
done
bash $R/tools/inflate_times.sh text 2000 | tail -1
bash $R/tools/inflate_times.sh small_files 30000 | tail -1
