#!/bin/bash
# round 4, call 1: search rework + row g + config 3 (small corpus) on the GPU
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4c1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_search.py -x -q -m gpu > $O/search_tests.log 2>&1; echo "search tests rc=$?" | tee -a $O/summary.txt
tail -5 $O/search_tests.log
VISRAG_TEST_CORPUS_PAGES=6400 timeout 600 python -m pytest tests/test_gpu_config3.py -x -q -m gpu -s > $O/config3_small.log 2>&1; echo "config3(6400) rc=$?" | tee -a $O/summary.txt
tail -8 $O/config3_small.log
timeout 900 python bench.py --corpus-pages 20000 --no-cpu-baseline > $O/bench_20k.json 2> $O/bench_20k.err; echo "bench rc=$?" | tee -a $O/summary.txt
tail -3 $O/bench_20k.err
python - <<'PY'
import json
j = json.load(open("gpurun_out/r4c1/bench_20k.json"))
print("value", j["value"], "ms", j["ms_per_step"])
for k in ("search", "search_filler", "search_templated"):
    s = j.get(k)
    if not s: continue
    print(k, s["index_kind"], "ms", s["ms_per_search"], "sweep", s["local_sweep_ms"], s["stages_ms"], s["certification"], s.get("ids_vs_fp64"), s.get("exact_pass_ms_per_8q"), s.get("embedding_stats"), s["error_model"], s["single_query"])
print(j["corpus_embed"], j.get("pil_pipeline"), j.get("pipelined"))
print({k: v for k, v in j["phases"].items()})
PY
