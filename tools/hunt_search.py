"""Bug hunt: N more seeds of tests/test_gpu_search.py::test_search_random_shapes_and_structure_vs_fp64 (random rows / queries /
width / k with duplicate rows, near-duplicate clusters and queries that are index rows, against an fp64 brute force).
    python tools/hunt_search.py 400        (round 3: 400 seeds, 0 failures)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.test_gpu_search as T
bad = []
for seed in range(24, 24 + int(sys.argv[1])):
    try:
        T.test_search_random_shapes_and_structure_vs_fp64(seed)
    except Exception as e:
        bad.append((seed, repr(e)[:300]))
print("failures:", len(bad))
for b in bad[:10]: print(b)
