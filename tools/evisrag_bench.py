"""EVisRAG-7B-shaped generation benchmark (visrag_amd.evisrag.bench_generate):
    python tools/evisrag_bench.py [n_images=5] [answer_tokens=64] [queries=2] [vision=1]   -> one JSON line
(VISRAG_HIP_LIB selects a tagged build of the library, like the other A/B tools.)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visrag_amd.evisrag import bench_generate  # noqa: E402

a = [int(x) for x in sys.argv[1:]]
n_images, answer, queries, vision = (a + [5, 64, 2, 1][len(a):])
r = bench_generate(n_images, answer, queries, 0, bool(vision))
r["lib"] = os.environ.get("VISRAG_HIP_LIB", "")
print(json.dumps(r))
