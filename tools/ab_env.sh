#!/bin/bash
# A/B an environment knob inside ONE GPU session (same box, same clocks): ab_env.sh VAR v1 v2 ...
var=$1; shift
for v in "$@"; do
  env $var=$v timeout 300 python bench.py --no-cpu-baseline --search-steps 3 --queries 64 --index-rows 20000 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$var=$v', d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['phases'].items()})"
done
