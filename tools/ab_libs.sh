#!/bin/bash
# In-model A/B of library builds on ONE box: interleaved rounds of tools/ab_encode.py.  usage: ab_libs.sh <outdir> <rounds> lib1 lib2 ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$1; R=$2; shift 2; mkdir -p $O
for r in $(seq 1 $R); do
  for L in "$@"; do
    VISRAG_HIP_LIB=$PWD/$L timeout 300 python tools/ab_encode.py 10 2 2>/dev/null | tail -1 >> $O/ab.log
  done
done
python - $O/ab.log <<'PY'
import json, sys, collections
d = collections.defaultdict(list)
for l in open(sys.argv[1]):
    try: j = json.loads(l)
    except Exception: continue
    d[j["lib"].split("/")[-1]].append(j["best"])
for k, v in d.items():
    ms = sorted(x["ms_per_step"] for x in v)
    ph = {p: round(min(x["phases_ms"][p] for x in v), 3) for p in v[0]["phases_ms"]}
    print(k, "ms/step min %.3f med %.3f" % (ms[0], ms[len(ms)//2]), ph)
PY
