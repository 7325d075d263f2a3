import sys, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.microbench import bench_gemm
bench_gemm(8192, 8192, 8192, 0, 9)
vs = [int(x) for x in sys.argv[1:]] or [9, 12]
for (M, N, K, epi) in ((8192, 8192, 8192, 0), (32768, 3456, 1152, 0), (32768, 4352, 1152, 1), (32768, 4352, 1152, 0), (2176, 6912, 2304, 0), (2176, 2304, 2304, 2)):
    for v in vs:
        print(json.dumps(bench_gemm(M, N, K, epi, v)))
