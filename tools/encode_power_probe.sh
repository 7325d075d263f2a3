#!/bin/bash
# Package power / shader clock while the encode loop runs, per library build:  encode_power_probe.sh lib1.so lib2.so ...
R=$GRAFT_REPO_ROOT
for L in "$@"; do
  VISRAG_HIP_LIB=$R/$L python $R/tools/encode_only.py 150 > /tmp/enc_p.log 2>&1 &
  PID=$!
  sleep 9
  for i in 1 2 3 4 5; do
    kill -0 $PID 2>/dev/null || break
    echo "$L $(rocm-smi --showclocks --showpower 2>/dev/null | grep -i 'sclk\|Package Power' | sed 's/GPU\[\([0-9]*\)\][^:]*: /g\1 /' | tr '\n' ';' | cut -c1-160)"
    sleep 0.5
  done
  wait $PID; tail -1 /tmp/enc_p.log | cut -c1-200
done
