"""Main-loop ablations of the 256^2 GEMM (variants 20..26 of vr_op_gemm; timing only) next to the shipped kernel (9)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.microbench import bench_gemm
bench_gemm(8192, 8192, 8192, 0, 9)
for shape in ((8192, 8192, 8192), (32768, 4352, 1152)):
    for v in [int(x) for x in sys.argv[1:]] or (9, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29):
        print(json.dumps(bench_gemm(*shape, 0, v)))
