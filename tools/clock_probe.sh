#!/bin/bash
# Shader clock / power while the encode loop runs (is the chip power-capped under this load?)
R=$GRAFT_REPO_ROOT
python $R/tools/encode_only.py 400 > /tmp/enc.log 2>&1 &
PID=$!
for i in $(seq 1 40); do
  sleep 0.7
  echo "t=$i $(rocm-smi --showclocks --showpower 2>/dev/null | grep -i 'sclk\|Package Power' | sed 's/GPU\[\([0-9]*\)\][^:]*: /g\1 /' | tr '\n' ';' | cut -c1-300)"
  kill -0 $PID 2>/dev/null || break
done
wait $PID; tail -1 /tmp/enc.log
