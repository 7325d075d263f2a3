#!/bin/bash
# round 4, call 5: full -m gpu suite (config 3 at 20k pages) + the N = 2 bench path on one GPU (ranks share the device, gloo)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4c5; mkdir -p $O
VISRAG_TEST_CORPUS_PAGES=20000 timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?" | tee -a $O/summary.txt
tail -4 $O/gpu_tests.log
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 1 --corpus-pages 6400 --no-cpu-baseline > $O/bench_n2.json 2> $O/bench_n2.err; echo "bench n2 rc=$?" | tee -a $O/summary.txt
tail -3 $O/bench_n2.err
python - <<'PY'
import json
try:
    j = json.loads(open("gpurun_out/r4c5/bench_n2.json").read().strip().splitlines()[-1])
    print("n2 value", j["value"], j["n_gpus"], j["search"]["index_kind"], j["search"]["ms_per_search"], j["search"]["exchange"], j["search"]["ids_vs_fp64"], j["corpus_embed"])
except Exception as e:
    print("parse failed", e)
PY
