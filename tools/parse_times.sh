#!/bin/bash
# GPU-box helper: per-phase cycle counts of lz_parse (instrumented build, ZWZ_PARSE_EXP=16).  usage: tools/inflate_times.sh [workload] [files]
R=$GRAFT_REPO_ROOT; W=${1:-text}; F=${2:-2000}
cd $R/parallel-data-compression-and-decompression_amd
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 -DZWZ_PARSE_EXP=16 -shared -o libzwz_hip.so csrc/zwz_kernels.hip csrc/zwz_band.hip csrc/zwz_plan.hip csrc/zwz_api.cpp csrc/zwz_host.cpp csrc/zwz_pipeline.cpp 2> /tmp/inf_build.err || { echo build failed; tail -20 /tmp/inf_build.err; exit 1; }
cd $R && ZWZ_PARSE_TIMES=1 timeout -k 10 300 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --workload $W --files $F 2>&1 | grep -E "ZWZ_PARSE_TIMES" | tail -2
