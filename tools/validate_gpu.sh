#!/bin/bash
# GPU-box helper: the parity suite, then the default bench line (both headline workloads + CPU baseline).
# usage: tools/validate_gpu.sh <tag>
TAG=${1:-r02}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_$TAG.log 2>&1; rc=$?
tail -15 gpurun_out/gpu_tests_$TAG.log
[ $rc -ne 0 ] && { echo "GPU TESTS FAILED rc=$rc"; exit $rc; }
timeout -k 10 600 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; rc=$?
tail -c 1500 gpurun_out/bench_$TAG.err
python - <<PY
import json
d = json.loads(open("gpurun_out/bench_$TAG.json").read().strip().splitlines()[-1])
for nm, r in (("random", d), ("text", d.get("text", {}))):
    print(nm, r.get("value"), "c", r.get("compress_GBps"), "d", r.get("decompress_GBps"), r.get("verified"), r.get("stage_ms_per_pass"))
    print("   roofline", r.get("roofline"))
    cb = r.get("cpu_baseline") or {}
    print("   cpu", cb.get("value"), cb.get("kind"), cb.get("ranks"), cb.get("host"))
PY
exit $rc
