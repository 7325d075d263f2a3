"""Helpers for the -m gpu tests: call the op-level C-ABI entry points on torch containers."""
import ctypes as C

import torch

from visrag_amd import _lib


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def pad_rows(t, mult=256):
    r = (t.shape[0] + mult - 1) // mult * mult
    if r == t.shape[0]:
        return t.contiguous()
    out = torch.zeros((r,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    out[: t.shape[0]] = t
    return out


def op_gemm(A, W, epi, bias=None, resid=None, alpha=1.0, out_dtype=torch.bfloat16, out_cols=None,
            rope_pos=None, rope_table=None, rope_cols=0, variant=0):
    """A [M,K] bf16, W [N,K] bf16 (cuda) -> out [M, out_cols]."""
    lib = _lib.load()
    M, K = A.shape
    N = W.shape[0]
    Ap = pad_rows(A)
    out_cols = out_cols or N
    out = torch.zeros((Ap.shape[0], out_cols), dtype=out_dtype, device=A.device)
    if resid is not None:
        resid = pad_rows(resid.to(torch.float32))
    _lib.check(lib.vr_op_gemm(A.device.index or 0, P(Ap), K, P(W.contiguous()), K, M, N, K, epi, P(bias), P(resid),
                              float(alpha), P(out), out_cols, P(rope_pos), P(rope_table), rope_cols, variant, None),
               "vr_op_gemm")
    torch.cuda.synchronize()
    return out[:M]


def op_norm(kind, x, w, b, eps, ldo=None):
    lib = _lib.load()
    rows, dim = x.shape
    ldo = ldo or dim
    out = torch.full((rows, ldo), 7.0, dtype=torch.bfloat16, device=x.device)
    _lib.check(lib.vr_op_norm(x.device.index or 0, kind, P(x.contiguous()), rows, dim, P(w), P(b), float(eps), P(out),
                              ldo, None), "vr_op_norm")
    torch.cuda.synchronize()
    return out


def op_attention(q, k, v, cu_q, cu_kv, heads, hd, max_q, causal, q_shared, scale, rows_out):
    lib = _lib.load()
    out = torch.zeros((rows_out, heads * hd), dtype=torch.bfloat16, device=q.device)
    B = cu_kv.numel() - 1
    _lib.check(lib.vr_op_attention(q.device.index or 0, P(q), q.stride(0), P(k), k.stride(0), P(v), v.stride(0),
                                   P(out), out.stride(0), P(cu_q), P(cu_kv), B, heads, hd, max_q, int(causal),
                                   int(q_shared), float(scale), None), "vr_op_attention")
    torch.cuda.synchronize()
    return out
