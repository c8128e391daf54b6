#!/bin/bash
# GPU-box helper: codec parity tests (fast subset), then both bench workloads without the CPU baseline.  usage: tools/quick_gpu.sh <tag>
TAG=${1:-q}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_codec.py -m gpu -x -q > gpurun_out/gpu_quick_$TAG.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_quick_$TAG.log
[ $rc -ne 0 ] && { echo "GPU TESTS FAILED rc=$rc"; tail -40 gpurun_out/gpu_quick_$TAG.log; exit $rc; }
timeout -k 10 500 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --oracle-sample 128 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; rc=$?
tail -c 600 gpurun_out/bench_$TAG.err
python - <<PY
import json
d = json.loads(open("gpurun_out/bench_$TAG.json").read().strip().splitlines()[-1])
for nm, r in (("random", d), ("text", d.get("text", {}))):
    print(nm, r.get("value"), "c", r.get("compress_GBps"), "d", r.get("decompress_GBps"), "ok", r.get("verified", {}).get("ok"), r.get("stage_ms_per_pass"))
PY
exit $rc
