#!/bin/bash
# GPU-box runner: ONE script for everything that is run on the MI355X box (replaces round 3's per-phase scripts).
#   tools/gpu.sh tests  <tag> [pytest args ...]        pytest -m gpu (default: the whole suite), log in gpurun_out/tests_<tag>.log
#   tools/gpu.sh bench  <tag> [bench.py args ...]      one bench.py run, JSON line in gpurun_out/bench_<tag>.json, stage split printed
#   tools/gpu.sh trace  <tag> <workload> <files>       rocprofv3 --kernel-trace of one bench pass: per-kernel durations
#   tools/gpu.sh prof   <tag> <workload> <files>       kernel trace + the PMC passes (each in its own run), summary + traffic.json
#   tools/gpu.sh times  <kernel> <workload> <files> [bits]   per-phase cycle stamps of band|match|parse|enc|inf: an instrumented build
#                                                      (-DZWZ_<K>_EXP=16|bits) goes to libzwz_hip_exp.so and is loaded through ZWZ_LIB;
#                                                      the product library libzwz_hip.so is NEVER rebuilt or replaced by this script
#   tools/gpu.sh variant <tag> "<flags>" [bench args]  a build with other compile-time constants (libzwz_hip_exp.so too) and one bench run on it
#   tools/gpu.sh soak   <tag> [chunks] [seed]          tools/soak_gpu.py
#   tools/gpu.sh refresh <tag>                         the evidence behind DESIGN.md's tables: prof of both headline workloads at full
#                                                      size -> profiles/<tag>_*_rocprofv3_summary.txt + profiles/traffic_*.json, then the default bench line
# Steps inside one call are joined with &&: nothing runs on the GPU after a step failed or timed out.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
PKG=$R/parallel-data-compression-and-decompression_amd
mkdir -p $R/gpurun_out
cmd=$1; shift

kernel_table() {   # <trace dir>: per-kernel calls / max / median of the zwz kernels
python3 - "$1" <<'PY'
import csv, glob, collections, sys
dur = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "zwz" in r["Kernel_Name"]: dur[r["Kernel_Name"].split("(")[0].replace("zwz::", "")].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(dur.items(), key=lambda kv: -max(kv[1])):
    v2 = sorted(v)
    print("%-28s calls=%d max_ms=%.3f median_ms=%.3f" % (k, len(v), max(v) / 1e6, v2[len(v2) // 2] / 1e6))
PY
}

bench_table() {    # <json file>: the line's headline figures
python3 - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
rows = [(d["config"]["workload"][:40], d)] + [(k, d[k]) for k in ("text", "random") if isinstance(d.get(k), dict)]
for nm, r in rows:
    print(nm, "value", r.get("value"), "c", r.get("compress_GBps"), "d", r.get("decompress_GBps"), "ok", r.get("verified", {}).get("ok"), r.get("stage_ms_per_pass"))
    if r.get("cpu_baseline"): print("   cpu", r["cpu_baseline"].get("value"), r["cpu_baseline"].get("compress_GBps"), r["cpu_baseline"].get("decompress_GBps"), r["cpu_baseline"].get("note"))
for k, v in (d.get("e2e") or {}).items():
    if isinstance(v, dict): print("e2e", k, v.get("filesystem"), "compress", v["compress"].get("banner_s"), "s", v["compress"].get("GBps"), "GB/s steady", v["compress"].get("steady_GBps"), "| decompress", v["decompress"].get("banner_s"), "s", v["decompress"].get("GBps"), "GB/s steady", v["decompress"].get("steady_GBps"), "ok", v.get("ok"))
    else: print("e2e", k, v)
PY
}

case $cmd in
tests)
  TAG=${1:-t}; shift
  [ $# -eq 0 ] && set -- tests
  cd $R && timeout -k 10 1100 python -m pytest "$@" -m gpu -x -q > gpurun_out/tests_$TAG.log 2>&1; rc=$?
  tail -4 gpurun_out/tests_$TAG.log
  [ $rc -ne 0 ] && { echo "GPU TESTS FAILED rc=$rc"; tail -60 gpurun_out/tests_$TAG.log; }
  exit $rc ;;
bench)
  TAG=${1:-b}; shift
  cd $R && timeout -k 10 1100 python bench.py "$@" > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; rc=$?
  tail -c 800 gpurun_out/bench_$TAG.err
  [ $rc -eq 0 ] && bench_table gpurun_out/bench_$TAG.json
  exit $rc ;;
trace)
  TAG=${1:-t}; W=${2:-text}; F=${3:-2000}
  O=$R/gpurun_out/trace_${TAG}_$W; rm -rf $O; mkdir -p $O
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --workload $W --files $F > $O/trace.log 2>&1 || { echo "trace failed"; tail -20 $O/trace.log; exit 1; }
  kernel_table $O/trace; tail -c 300 $O/trace.log ;;
prof)
  TAG=${1:-p}; W=${2:-text}; F=${3:-10000}
  # (gpurun MERGES gpurun_out/ back: a directory name used twice would hold both runs' CSVs and the summary would count launches twice --
  # so on the CPU side, delete gpurun_out/prof_<tag>_<workload> before a second run under the same tag)
  O=$R/gpurun_out/prof_${TAG}_$W; rm -rf $O; mkdir -p $O
  cd /tmp && export TMPDIR=/tmp
  ARGS="$R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --workload $W --files $F"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $ARGS > $O/trace.log 2>&1 && \
  timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc1 -- python3 $ARGS > $O/pmc1.log 2>&1 && \
  timeout -k 10 400 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $O/pmc2 -- python3 $ARGS > $O/pmc2.log 2>&1 && \
  timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc3 -- python3 $ARGS > $O/pmc3.log 2>&1 && \
  timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc4 -- python3 $ARGS > $O/pmc4.log 2>&1 || { echo "prof failed"; tail -5 $O/*.log; exit 1; }
  python3 $R/tools/prof_summary.py $O > $O/summary.stdout && grep -A24 "kernel stats" $O/summary.txt | cut -c1-150 && grep -A20 "HBM traffic per launch" $O/summary.txt | cut -c1-220 ;;
times)
  K=${1:-band}; W=${2:-text}; F=${3:-2000}; X=$((16 | ${4:-0}))
  case $K in band) D=ZWZ_BAND_EXP; E=ZWZ_BAND_TIMES;; match) D=ZWZ_MATCH_EXP; E=ZWZ_MATCH_TIMES;; parse) D=ZWZ_PARSE_EXP; E=ZWZ_PARSE_TIMES;;
             enc) D=ZWZ_ENC_EXP; E=ZWZ_ENC_TIMES;; inf) D=ZWZ_INF_EXP; E=ZWZ_INF_TIMES;; lazy) D=ZWZ_LAZY_EXP; E=ZWZ_LAZY_TIMES;; *) echo "times: band|match|parse|enc|inf|lazy"; exit 2;; esac
  make -C $PKG EXP_FLAGS="-D$D=$X" libzwz_hip_exp.so > $R/gpurun_out/exp_build.log 2>&1 || { echo "experiment build failed"; tail -30 $R/gpurun_out/exp_build.log; exit 1; }
  cd $R && ZWZ_LIB=$PKG/libzwz_hip_exp.so env $E=1 timeout -k 10 400 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --workload $W --files $F 2>&1 | grep -E "$E|EXPERIMENT|stage_ms" | cut -c1-700 ;;
variant)   # <tag> "<compiler flags>" [bench.py args ...]: a build with other constants (e.g. -DZWZ_INF_WAVES=4) as libzwz_hip_exp.so, one bench run on it
  TAG=${1:-v}; FL=$2; shift; shift
  make -C $PKG EXP_FLAGS="$FL" libzwz_hip_exp.so > $R/gpurun_out/exp_build.log 2>&1 || { echo "variant build failed"; tail -30 $R/gpurun_out/exp_build.log; exit 1; }
  cd $R && ZWZ_LIB=$PKG/libzwz_hip_exp.so timeout -k 10 600 python3 bench.py "$@" > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; rc=$?
  tail -c 400 gpurun_out/bench_$TAG.err; echo "variant $FL: rc=$rc"; [ -s gpurun_out/bench_$TAG.json ] && bench_table gpurun_out/bench_$TAG.json
  # rc 3 = verification failed: expected of a timing-only experiment, the stage times still stand.  Anything else -- a time-out (124 / 137), a
  # crash, a GPU fault -- is a failure: a caller that chains variants with && must not launch the next one on this box.
  [ $rc -eq 0 ] || [ $rc -eq 3 ] && exit 0
  tail -40 gpurun_out/bench_$TAG.err; exit $rc ;;
soak)
  TAG=${1:-s}; N=${2:-40000}; SEED=${3:-3}
  cd $R && timeout -k 10 900 python tools/soak_gpu.py $N $SEED > gpurun_out/soak_$TAG.log 2>&1 || { echo "soak failed"; tail -20 gpurun_out/soak_$TAG.log; exit 1; }
  tail -2 gpurun_out/soak_$TAG.log ;;
refresh)
  TAG=${1:-r04}
  bash $R/tools/gpu.sh prof $TAG random 10000 > $R/gpurun_out/prof_${TAG}_random.log 2>&1 && \
  bash $R/tools/gpu.sh prof $TAG text 10000 > $R/gpurun_out/prof_${TAG}_text.log 2>&1 && \
  cd $R && for w in random text; do cp gpurun_out/prof_${TAG}_$w/traffic.json gpurun_out/${TAG}_traffic_$w.json && cp gpurun_out/prof_${TAG}_$w/summary.txt gpurun_out/${TAG}_${w}_rocprofv3_summary.txt && \
    cp gpurun_out/${TAG}_traffic_$w.json profiles/traffic_$w.json; done && \
  # (profiles/ on the GPU box is a scratch copy: bench.py below reads the fresh traffic files from it; on the CPU side copy
  #  gpurun_out/<tag>_traffic_*.json and gpurun_out/<tag>_*_rocprofv3_summary.txt into profiles/ and commit them)
  timeout -k 10 900 python bench.py > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err; rc=$?
  tail -c 300 gpurun_out/bench_${TAG}.err; grep -A12 "kernel stats" gpurun_out/${TAG}_text_rocprofv3_summary.txt | cut -c1-150
  [ $rc -eq 0 ] && bench_table gpurun_out/bench_${TAG}.json
  exit $rc ;;
*)
  sed -n 2,16p $0; exit 2 ;;
esac
