#!/bin/bash
# GPU-box helper: parity tests, then both bench workloads (no CPU baseline), printing the stage split.
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; tail -3 gpurun_out/gpu_tests.log
for w in random text; do
  timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload $w > gpurun_out/bench_$w.log 2>&1
  tail -1 gpurun_out/bench_$w.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['value'], 'c', d['compress_GBps'], 'd', d['decompress_GBps'], d['payload_ratio'], d['roundtrip_property_ok'], d['stage_ms_per_pass'])" || tail -5 gpurun_out/bench_$w.log
done
