#!/bin/bash
# round 4, call 4: new tests (streamer text path, small-nq search, long-prompt generator), search stages, short bench
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4c4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_text_path.py tests/test_gpu_search.py tests/test_gpu_evisrag_long.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
tail -6 $O/tests.log
python tools/search_diag.py 100000 2304 1,16,256,1000 2>/dev/null | tee $O/search_stages.txt
timeout 900 python bench.py --corpus-pages 3200 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
tail -3 $O/bench.err
python - <<'PY'
import json
j = json.load(open("gpurun_out/r4c4/bench.json"))
print("value", j["value"], "ms", j["ms_per_step"], "pil", j.get("pil_pipeline"), "pipelined", j.get("pipelined"))
print("search", j["search"]["ms_per_search"], j["search"]["single_query"], j["search"]["query_encode_per_sec"])
print("filler", j["search_filler"]["ms_per_search"], j["search_filler"]["stages_ms"], j["search_filler"]["single_query"])
g = j.get("evisrag_generate") or {}
print("gen", {k: g.get(k) for k in ("decode_ms_per_token", "queries_per_s", "prefill_ms", "a4_pages")}, (g.get("end_to_end") or {}))
print(j.get("evisrag_error"), j.get("extras_error"))
print({k: v for k, v in j["phases"].items() if k.startswith("dec")})
PY
