#!/bin/bash
# GPU-box helper: the evidence behind DESIGN.md's tables -- rocprofv3 kernel stats + PMC passes for both headline
# workloads at full size, then the default bench line (both workloads + CPU baseline).  Results land in gpurun_out/;
# the summaries are copied into profiles/ under the round's tag.   usage: tools/refresh_profiles.sh <tag>
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
timeout -k 10 600 $R/tools/prof_gpu.sh random 10000 $TAG > $R/gpurun_out/prof_${TAG}_random.log 2>&1 && \
timeout -k 10 900 $R/tools/prof_gpu.sh text 10000 $TAG > $R/gpurun_out/prof_${TAG}_text.log 2>&1 && \
cd $R && cp gpurun_out/prof_${TAG}_random/traffic.json gpurun_out/${TAG}_traffic_random.json && cp gpurun_out/prof_${TAG}_text/traffic.json gpurun_out/${TAG}_traffic_text.json && \
cp gpurun_out/prof_${TAG}_random/summary.txt gpurun_out/${TAG}_random_rocprofv3_summary.txt && cp gpurun_out/prof_${TAG}_text/summary.txt gpurun_out/${TAG}_text_rocprofv3_summary.txt && \
cp gpurun_out/${TAG}_traffic_random.json profiles/traffic_random.json && cp gpurun_out/${TAG}_traffic_text.json profiles/traffic_text.json && \
timeout -k 10 600 python bench.py > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err
tail -c 300 gpurun_out/bench_${TAG}.err; grep "kernel stats" -A9 gpurun_out/${TAG}_text_rocprofv3_summary.txt | cut -c1-150; grep -A12 "HBM traffic per launch" gpurun_out/${TAG}_text_rocprofv3_summary.txt | cut -c1-200
