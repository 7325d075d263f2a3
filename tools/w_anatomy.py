"""Tile anatomy of the one-wave-per-SIMD GEMM (variant 12) from in-kernel timestamps.
Needs a tagged build:  python -m visrag_amd.build --tag wtm -DVR_W_TIMING
                       VISRAG_HIP_LIB=visrag_amd/libvisrag_hip_wtm.so python tools/w_anatomy.py
Wave 0 of every workgroup records s_memrealtime (100 MHz) at entry, after the prologue, after the
K-loop, after the epilogue's last instruction and after its stores are acknowledged, plus HW_ID /
XCC_ID, so launches can be ordered per CU."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.gpu_util import P
from visrag_amd import _lib
lib = _lib.load()
s = torch.cuda.current_stream().cuda_stream
shapes = [(32768, 3456, 1152, 0), (32768, 4352, 1152, 1)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split(',')) for a in sys.argv[1:]]
for sh in shapes:
    M, N, K, epi = sh[:4]
    variant, BN = (13, 192) if (epi == 3 and N % 192 == 0) else (12, 256)      # the product's choice for the N = 1152 residual GEMMs
    BM = 256
    if len(sh) > 4:
        variant, BN = sh[4], (192 if sh[4] in (13, 14) else 256)
    BM = 128 if variant in (14, 15) else 256
    Np = (N + BN - 1) // BN * BN
    A = torch.randn(((M + 255) // 256 * 256, K), device="cuda").to(torch.bfloat16)
    W = (torch.randn((Np, K), device="cuda") * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    out = torch.zeros(((M + 255) // 256 * 256, N), device="cuda", dtype=torch.float32 if epi in (2, 3) else torch.bfloat16)
    resid = out if epi == 3 else None
    tiles = ((M + BM - 1) // BM) * (Np // BN)
    dbg = torch.zeros((tiles, 16), dtype=torch.int64, device="cuda")
    for it in range(3):
        _lib.check(lib.vr_op_gemm(0, P(A), K, P(W), K, M, N, K, epi, P(bias), P(resid), 0.0 if epi == 3 else 1.0, P(out), N, None, P(dbg), 0, variant, s))
    torch.cuda.synchronize()
    d = dbg.cpu().numpy()
    t = d[:, :5].astype(np.float64) * 0.01          # us
    t0 = t[:, 0].min()
    cu = (d[:, 6] & 0xF) * 1000 + ((d[:, 5] >> 13) & 0x7) * 100 + ((d[:, 5] >> 8) & 0xF)   # xcc, se, cu
    pro, loop, epi_i, epi_w = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3]
    gaps = []
    for c in np.unique(cu):
        idx = np.where(cu == c)[0]
        idx = idx[np.argsort(t[idx, 0])]
        gaps += list(t[idx[1:], 0] - t[idx[:-1], 4])
    gaps = np.array(gaps if gaps else [0.0])
    print(json.dumps({"shape": [M, N, K], "epi": epi, "tiles": tiles, "cus": int(len(np.unique(cu))),
                      "kernel_us": round(float(t[:, 4].max() - t0), 1),
                      "prologue_us": [round(float(np.median(pro)), 2), round(float(pro.max()), 2)],
                      "kloop_us": [round(float(np.median(loop)), 2), round(float(loop.max()), 2)],
                      "epilogue_issue_us": [round(float(np.median(epi_i)), 2), round(float(epi_i.max()), 2)],
                      "store_ack_us": [round(float(np.median(epi_w)), 2), round(float(epi_w.max()), 2)],
                      "gap_between_wgs_on_a_cu_us": [round(float(np.median(gaps)), 2), round(float(gaps.max()), 2)],
                      "kloop_mhz": round(float(np.median(d[:, 7] / np.maximum(loop, 1e-3))), 0), "kstep_clocks": round(float(np.median(d[:, 7])) / (K // 64), 0),
                      "first_start_spread_us": round(float(np.sort(t[:, 0])[min(255, tiles - 1)] - t0), 2)}))
