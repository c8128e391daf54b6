#!/bin/bash
# GPU-box helper: rocprofv3 kernel stats + PMC passes of one bench workload.  usage: tools_prof.sh <workload> <files> <tag>
W=${1:-random}; F=${2:-2000}; TAG=${3:-r01}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_${TAG}_$W; mkdir -p $O
ARGS="$R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --workload $W --files $F"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $ARGS > $O/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc1 -- python3 $ARGS > $O/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $O/pmc2 -- python3 $ARGS > $O/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc3 -- python3 $ARGS > $O/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc4 -- python3 $ARGS > $O/pmc4.log 2>&1
find $O -name "*.csv" | head -20
python3 $R/tools/prof_summary.py $O
