"""A/B of GEMM variants on the model's shapes: gemm_ab.py v1 v2 ..."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.microbench import bench_gemm
vs = [int(x) for x in sys.argv[1:]] or [9, 11]
bench_gemm(8192, 8192, 8192, 0, 9)
shapes = [(8192, 8192, 8192, 0), (32768, 3456, 1152, 0), (32768, 4352, 1152, 1), (2176, 6912, 2304, 0), (2176, 11520, 2304, 4)]
for sh in shapes:
    print(json.dumps([bench_gemm(*sh, v) for v in vs]))
