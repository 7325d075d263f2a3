"""EVisRAG-7B-shaped generation benchmark (visrag_amd.evisrag.bench_generate):
    python tools/evisrag_bench.py [n_images=5] [answer_tokens=64] [queries=2]   -> one JSON line"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visrag_amd.evisrag import bench_generate  # noqa: E402

a = [int(x) for x in sys.argv[1:]]
print(json.dumps(bench_generate(*(a + [5, 64, 2][len(a):]))))
