"""Energy per call of single kernels: run one op back to back for a few seconds while a thread samples
the package power (rocm-smi), report sustained rate, mean watts, mean sclk and joules per call.
The chip is capped at 1400 W and the encode loop sits at that cap, so at model level a kernel's ENERGY
decides the step time as much as its isolated duration does.
    python tools/op_energy.py gemm9 gemm12 vendor attn ln fc1_9 fc1_12 fc2_7 ..."""
import json, os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.gpu_util import P
from visrag_amd import _lib
lib = _lib.load()
s = torch.cuda.current_stream().cuda_stream
SECS = float(os.environ.get("OP_SECS", "5"))


def sampler(stop, out):
    while not stop.is_set():
        try:
            t = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            w = re.search(r"Package Power \(W\): ([0-9.]+)", t)
            c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", t)
            if w and c:
                out.append((time.time(), float(w.group(1)), float(c.group(1))))
        except Exception:
            pass
        time.sleep(0.15)


def gemm_op(M, N, K, epi, variant):
    Np = (N + 255) // 256 * 256
    A = torch.randn((M, K), device="cuda").to(torch.bfloat16)
    W = (torch.randn((Np, K), device="cuda") * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    odt = torch.float32 if epi in (2, 3) else torch.bfloat16
    out = torch.zeros((M, N), device="cuda", dtype=odt)
    resid = out if epi == 3 else None
    Wt = W[:N].t()
    if variant == "vendor":
        return (lambda: torch.matmul(A, Wt, out=out)), 2.0 * M * N * K
    return (lambda: _lib.check(lib.vr_op_gemm(0, P(A), K, P(W), K, M, N, K, epi, P(bias), P(resid), 0.0 if epi == 3 else 1.0,
                                              P(out), N, None, None, 0, int(variant), s))), 2.0 * M * N * K


def gemm_op_swiglu(M, N, K, variant):
    A = torch.randn((M, K), device="cuda").to(torch.bfloat16)
    W = (torch.randn((N, K), device="cuda") * 0.05).to(torch.bfloat16)
    out = torch.zeros((M, N // 2), device="cuda", dtype=torch.bfloat16)
    return (lambda: _lib.check(lib.vr_op_gemm(0, P(A), K, P(W), K, M, N, K, 4, None, None, 1.0, P(out), N // 2, None, None, 0,
                                              int(variant), s))), 2.0 * M * N * K


def make(name):
    if name.startswith("gemm") or name == "vendor":          # ViT qkv
        return gemm_op(32768, 3456, 1152, 0, name[4:] if name != "vendor" else "vendor")
    if name.startswith("fc1_"):
        return gemm_op(32768, 4352, 1152, 1, name[4:])
    if name.startswith("fc2_"):
        return gemm_op(32768, 1152, 4352, 3, name[4:])
    if name.startswith("proj_"):
        return gemm_op(32768, 1152, 1152, 3, name[5:])
    if name.startswith("gu_"):            # decoder gate/up, SwiGLU
        return gemm_op_swiglu(2176, 11520, 2304, name[3:])
    if name.startswith("dqkv_"):          # decoder qkv (plain bf16 epilogue stands in for RoPE)
        return gemm_op(2176, 6912, 2304, 0, name[5:])
    if name.startswith("seq_"):           # ViT block front: LayerNorm -> qkv GEMM (variant) -> attention
        f1, _ = make("ln"); f2, fl2 = gemm_op(32768, 3456, 1152, 0, name[4:]); f3, fl3 = make("attn")
        return (lambda: (f1(), f2(), f3())), fl2 + fl3
    if name.startswith("mlp_"):           # LayerNorm -> fc1 (variant a) -> fc2 (variant b), name mlp_a_b
        va, vb = name[4:].split("_")
        f1, _ = make("ln"); f2, fl2 = gemm_op(32768, 4352, 1152, 1, va); f3, fl3 = gemm_op(32768, 1152, 4352, 3, vb)
        return (lambda: (f1(), f2(), f3())), fl2 + fl3
    if name == "encode":                  # one full encode step (32 pages), energy per step
        from PIL import Image
        from visrag_amd.config import full_config
        from visrag_amd.engine import HipEncoder
        from visrag_amd.preprocess import prepare_batch
        from visrag_amd.synth import iter_synth_weights, synth_pages
        from visrag_amd.tokenizer import StandInTokenizer
        cfg = full_config(); B = 32
        enc = HipEncoder(cfg, max_images=B, max_tokens=4096, max_seqs=64)
        enc.load_state_dict(iter_synth_weights(cfg, 0, device="cuda"))
        tok = StandInTokenizer(cfg.vocab_size)
        pages = synth_pages(B, size=448, seed=0)
        items = prepare_batch([""] * B, [Image.fromarray(p) for p in pages], tok, cfg, 2048)
        dev = [torch.from_numpy(p).cuda() for p in pages]
        out = torch.empty((B, cfg.hidden_size), dtype=torch.float32, device="cuda")
        return (lambda: enc.encode_items(items, device_slices=dev, out=out)), cfg.flops_page(1024, len(items[0].input_ids)) * B
    if name == "attn":
        B, N, heads, hd = 32, 1024, 16, 72
        Wd = heads * hd; ld = (3 * Wd + 127) // 128 * 128
        qkv = torch.randn((B * N, ld), device="cuda").to(torch.bfloat16)
        out = torch.zeros((B * N, (Wd + 127) // 128 * 128), dtype=torch.bfloat16, device="cuda")
        cu = (torch.arange(B + 1, dtype=torch.int32) * N).cuda()
        return (lambda: _lib.check(lib.vr_op_attention(0, P(qkv), ld, qkv.data_ptr() + Wd * 2, ld, qkv.data_ptr() + 2 * Wd * 2, ld,
                                                       P(out), out.stride(0), P(cu), P(cu), B, heads, hd, N, 0, 0, hd ** -0.5, s))), \
            4.0 * B * N * N * Wd
    if name == "ln":
        M, K = 32768, 1152
        x = torch.randn((M, K), device="cuda"); w = torch.ones(K, device="cuda"); b = torch.zeros(K, device="cuda")
        o = torch.empty((M, K), device="cuda", dtype=torch.bfloat16)
        return (lambda: _lib.check(lib.vr_op_norm(0, 0, P(x), M, K, P(w), P(b), 1e-6, P(o), K, s))), 0.0
    raise SystemExit(f"unknown op {name}")


for name in sys.argv[1:]:
    fn, flops = make(name)
    REPS = 10 if name == "encode" else 100
    for _ in range(20): fn()
    torch.cuda.synchronize()
    stop, samples = threading.Event(), []
    th = threading.Thread(target=sampler, args=(stop, samples)); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < SECS:
        for _ in range(REPS): fn()
        torch.cuda.synchronize(); n += REPS
    dt = time.time() - t0
    stop.set(); th.join()
    ss = [x for x in samples if x[0] - t0 > 1.5]
    w = sum(x[1] for x in ss) / max(len(ss), 1); c = sum(x[2] for x in ss) / max(len(ss), 1)
    print(json.dumps({"op": name, "us_per_call": round(dt / n * 1e6, 1), "tflops": round(flops * n / dt / 1e12, 1), "watts": round(w),
                      "sclk_mhz": round(c), "mJ_per_call": round(w * dt / n * 1e3, 1), "samples": len(ss)}))
    time.sleep(1.0)
