#!/bin/bash
# GPU-box helper: shader clock and power while the text and the incompressible workloads run (is a VALU-bound kernel clock-limited?).
R=$GRAFT_REPO_ROOT; cd $R
for w in text random; do
  python3 bench.py --steps 40 --warmup 1 --no-cpu-baseline --workload $w --files 4000 > gpurun_out/clock_$w.json 2> gpurun_out/clock_$w.err &
  P=$!
  sleep 6
  for i in 1 2 3 4 5 6; do
    /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | tr '\n' ' '; echo
    sleep 0.7
  done
  wait $P
done
/opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo "(idle)"
