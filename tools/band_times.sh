#!/bin/bash
# GPU-box helper: per-phase cycle counts of lz_match_band (instrumented build, ZWZ_BAND_EXP=16) on the text workload.  usage: tools/band_times.sh [files] [extra exp bits]
R=$GRAFT_REPO_ROOT; F=${1:-2000}; X=$((16 | ${2:-0}))
cd $R/parallel-data-compression-and-decompression_amd
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 -DZWZ_BAND_EXP=$X -shared -o libzwz_hip.so csrc/zwz_kernels.hip csrc/zwz_band.hip csrc/zwz_plan.hip csrc/zwz_api.cpp csrc/zwz_host.cpp csrc/zwz_pipeline.cpp 2> /dev/null || { echo build failed; exit 1; }
cd $R && ZWZ_BAND_TIMES=1 timeout -k 10 300 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --workload text --files $F 2>&1 | grep -E "ZWZ_BAND_TIMES|stage_ms" | cut -c1-600
