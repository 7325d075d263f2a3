#!/bin/bash
# the -m gpu suite with the small corpus (the driver runs the 100k default) + a short bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/suite
VISRAG_TEST_CORPUS_PAGES=3200 timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/suite/tests.log 2>&1
tail -5 gpurun_out/suite/tests.log
timeout 600 python bench.py --corpus-pages 3200 --index-rows 100000 --steps 10 --warmup 3 > gpurun_out/suite/bench.json 2> gpurun_out/suite/bench.err
tail -c 600 gpurun_out/suite/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/suite/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['phases']['vit_attn'], d.get('pipelined'))
print('search', d['search']['ms_per_search'], d['search']['single_query'], 'templ', (d.get('search_templated') or {}).get('vs_filler_time'))
print('parity', (d.get('cpu_baseline') or {}).get('reference_parity'), d.get('reference_parity_error'))
print('box', d.get('box'))
PY
