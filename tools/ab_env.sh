#!/bin/bash
# In-model A/B of runtime switches on ONE box: interleaved rounds of tools/ab_encode.py under each environment setting.
#   usage: ab_env.sh <outdir> <rounds> "VR_LN_FUSE=0" "VR_LN_FUSE=1" ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$1; R=$2; shift 2; mkdir -p $O
for r in $(seq 1 $R); do
  for E in "$@"; do
    env $E timeout 300 python tools/ab_encode.py 10 2 2>/dev/null | tail -1 | sed "s#\"lib\": \"[^\"]*\"#\"lib\": \"$E\"#" >> $O/ab_env.log
  done
done
python - $O/ab_env.log <<'PY'
import json, sys, collections
d = collections.defaultdict(list)
for l in open(sys.argv[1]):
    try: j = json.loads(l)
    except Exception: continue
    d[j["lib"]].append(j)
for k, v in d.items():
    ms = sorted(x["best"]["ms_per_step"] for x in v)
    ph = {p: round(min(x["best"]["phases_ms"][p] for x in v), 3) for p in v[0]["best"]["phases_ms"]}
    print(k, "ms/step min %.3f med %.3f" % (ms[0], ms[len(ms)//2]), "checksum", v[0]["checksum"], ph)
PY
