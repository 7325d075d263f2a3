#!/bin/bash
# round 4, call 6: decoder o projection without split-K (in-model A/B), query-encode rate with the stacked hi|lo GEMM
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4c6; mkdir -p $O
bash tools/ab_libs.sh $O/ab 3 visrag_amd/libvisrag_hip.so visrag_amd/libvisrag_hip_onos.so 2>&1 | tee $O/ab_summary.txt
timeout 600 python bench.py --corpus-pages 0 --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
j = json.load(open("gpurun_out/r4c6/bench.json"))
print("value", j["value"], "q_enc/s", j["search"]["query_encode_per_sec"], "single", j["search"]["single_query"], j["search"]["ms_per_search"])
PY
