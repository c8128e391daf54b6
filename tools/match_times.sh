#!/bin/bash
# GPU-box helper: per-phase cycle counts of lz_match (instrumented build, ZWZ_MATCH_EXP=16) on the incompressible workload.  usage: tools/match_times.sh [files]
R=$GRAFT_REPO_ROOT; F=${1:-4000}
cd $R/parallel-data-compression-and-decompression_amd
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 -DZWZ_MATCH_EXP=16 -shared -o libzwz_hip.so csrc/zwz_kernels.hip csrc/zwz_band.hip csrc/zwz_plan.hip csrc/zwz_api.cpp csrc/zwz_host.cpp csrc/zwz_pipeline.cpp 2> /dev/null || { echo build failed; exit 1; }
cd $R && ZWZ_MATCH_TIMES=1 timeout -k 10 300 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --workload random --files $F 2>&1 | grep -E "ZWZ_MATCH_TIMES|stage_ms" | cut -c1-600
