#!/bin/bash
# A/B of library builds on the generator's decode step, on ONE box: usage  ab_gen.sh "" _tagA _tagB ""   (tags of
# `python -m visrag_amd.build --tag ...`; "" = the default build).  Prints decode ms/token (captured steps, host-driven
# steps) and prefill ms for a 7B-shaped language model with a short prompt.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for L in "$@"; do
  VISRAG_HIP_LIB=$PWD/visrag_amd/libvisrag_hip$L.so timeout 100 python tools/evisrag_bench.py 1 48 2 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib$L', j['decode_ms_per_token'], j['decode_ms_per_token_host_driven'], j['prefill_ms'])"
done
