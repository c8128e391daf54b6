#!/bin/bash
# GPU-box helper while working on the plan stage (zwz_plan.hip): codec parity tests, then stage times of the three workloads the
# stage matters for, each with the old lane-serial kernel (ZWZ_PLAN=serial) beside the default.
# usage: tools/plan_gpu.sh <tag>
TAG=${1:-p}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R && timeout -k 10 700 python -m pytest tests/test_gpu_codec.py -m gpu -x -q > gpurun_out/plan_$TAG.log 2>&1; rc=$?
tail -3 gpurun_out/plan_$TAG.log
[ $rc -ne 0 ] && { echo "GPU TESTS FAILED rc=$rc"; tail -60 gpurun_out/plan_$TAG.log; exit $rc; }
for mode in ${MODES:-wave}; do
  for w in "small_files --files 30000" "text --files 4000" "random --files 4000"; do
    ZWZ_PLAN=$mode timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload $w > gpurun_out/plan_${TAG}_${mode}.json 2> gpurun_out/plan_${TAG}_${mode}.err || { echo "bench failed ($mode, $w)"; tail -20 gpurun_out/plan_${TAG}_${mode}.err; exit 1; }
    python3 - "$mode" "$w" gpurun_out/plan_${TAG}_${mode}.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
d = d.get("text", d) if sys.argv[2].startswith("text") and "text" in d else d
print(sys.argv[1], sys.argv[2], "ok=%s" % d["verified"]["ok"], "compress=%.2f GB/s" % d["compress_GBps"], {k: v for k, v in d["stage_ms_per_pass"].items()})
PY
  done
done
