#!/bin/bash
# Shader clock and package power while ONE GEMM variant runs back to back for ~6 s:
#   bash tools/gemm_power_probe.sh 9 12 vendor        (variants of tools/gemm_loop.py)
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  python $R/tools/gemm_loop.py $v 6 > /tmp/gl_$v.log 2>&1 &
  PID=$!
  sleep 2.5
  for i in 1 2 3 4; do
    echo "variant=$v $(rocm-smi --showclocks --showpower 2>/dev/null | grep -i 'sclk\|Package Power' | sed 's/GPU\[\([0-9]*\)\][^:]*: /g\1 /' | tr '\n' ';' | cut -c1-200)"
    sleep 0.6
  done
  wait $PID; tail -1 /tmp/gl_$v.log
done
