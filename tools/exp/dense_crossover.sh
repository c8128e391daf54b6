#!/bin/bash
# Which search is faster for which chunks (round 5)?  lz_dense_list sends a chunk to sort + band when its sample says "chain-heavy";
# this runs the kernel-scope bench with every chunk on one path (ZWZ_MATCH=band | walk) for text-like chunks of 4 - 32 KB and image-like
# files of several mean sizes, and prints links + match per pass.   On the GPU box: bash tools/exp/dense_crossover.sh [text|image]
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
show() { python3 - "$@" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
s=d['stage_ms_per_pass']
print(*sys.argv[2:], 'value', d['value'], 'links+match ms', round(s['lz_links']+s['lz_match'],2), 'chunks', d['verified']['chunks'], 'ok', d['verified']['ok'])
PY
}
if [ "${1:-text}" = text ]; then
for fb in 4096 8192 16384 32768; do
  n=$((1310720000 / fb))
  for m in band walk; do
    ZWZ_MATCH=$m timeout -k 10 300 python bench.py --workload text --file-bytes $fb --files $n --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/cross_${fb}_$m.json 2> gpurun_out/cross_${fb}_$m.err || echo FAIL $fb $m
    show gpurun_out/cross_${fb}_$m.json text $fb $m
  done
done
else
for mean in 3000 6800 16000 40000; do
  n=$((1310720000 / mean))
  for m in band walk; do
    ZWZ_SMALL_MEAN=$mean ZWZ_MATCH=$m timeout -k 10 400 python bench.py --workload small_files --files $n --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/crossi_${mean}_$m.json 2> gpurun_out/crossi_${mean}_$m.err || echo FAIL $mean $m
    show gpurun_out/crossi_${mean}_$m.json image mean $mean $m
  done
done
fi
