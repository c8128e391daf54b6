#!/bin/bash
# GPU-box helper: parity + bench with each lz_links flavour
mkdir -p gpurun_out
for f in ${FLAVOURS:-default pair}; do
  ZWZ_LINKS=$f timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_$f.log 2>&1; echo "$f: $(tail -1 gpurun_out/gpu_tests_$f.log)"
  for w in random text; do
    ZWZ_LINKS=$f timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload $w > gpurun_out/bench_${f}_$w.log 2>&1
    tail -1 gpurun_out/bench_${f}_$w.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f $w', d['value'], 'c', d['compress_GBps'], d['roundtrip_property_ok'], d['stage_ms_per_pass'])" || tail -5 gpurun_out/bench_${f}_$w.log
  done
done
