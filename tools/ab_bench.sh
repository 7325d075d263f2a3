#!/bin/bash
# A/B the 256-tile GEMM variant inside the full model within ONE GPU session (same box, same clocks)
for v in "$@"; do
  VR_GEMM256=$v timeout 300 python bench.py --no-cpu-baseline --search-steps 3 --queries 64 --index-rows 20000 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant $v', d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['phases'].items()})"
done
