#!/bin/bash
# GPU-box helper: the evidence behind DESIGN.md's tables -- rocprofv3 kernel stats + PMC passes for both
# workloads, then the full bench lines (with the CPU baseline).  Results land in gpurun_out/; copy the
# summaries into profiles/ afterwards.   usage: tools/refresh_profiles.sh <tag>
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
timeout -k 10 500 $R/tools/prof_gpu.sh random 10000 $TAG > $R/gpurun_out/prof_${TAG}_random.log 2>&1 && \
timeout -k 10 500 $R/tools/prof_gpu.sh text 10000 $TAG > $R/gpurun_out/prof_${TAG}_text.log 2>&1 && \
cd $R && cp gpurun_out/prof_${TAG}_random/traffic.json profiles/traffic_random.json && cp gpurun_out/prof_${TAG}_text/traffic.json profiles/traffic_text.json && \
timeout -k 10 600 python bench.py > gpurun_out/bench_${TAG}_random.json 2> gpurun_out/bench_${TAG}_random.err && \
timeout -k 10 600 python bench.py --workload text > gpurun_out/bench_${TAG}_text.json 2> gpurun_out/bench_${TAG}_text.err
tail -c 600 gpurun_out/bench_${TAG}_random.json; tail -c 600 gpurun_out/bench_${TAG}_text.json
