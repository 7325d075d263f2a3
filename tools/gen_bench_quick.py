import sys, json; sys.path.insert(0, '/root/repo')
from visrag_amd.evisrag import bench_generate
print(json.dumps(bench_generate(5, 64, 2, 0)))
