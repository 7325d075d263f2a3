"""EVisRAG-7B-shaped generation (BASELINE config 5: top-5 retrieved pages fed to the generator, one query at a time
like src/evisrag/predict.py:128-149) on the language model of visrag_amd/evisrag.py — random weights of the
Qwen2.5-VL-7B architecture, image tokens as precomputed embedding rows (the vision tower is not built yet).

    python tools/evisrag_bench.py [n_images=5] [answer_tokens=128] [queries=3]

Prints one JSON line: prefill ms, decode ms/token and tokens/s, queries/s for this answer length, and the decode
step's weight-streaming rate against the HBM (every token reads all 15.2 GB of bf16 weights once)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle.qwen_gen_oracle import QwenGenConfig, weight_specs  # noqa: E402  (shapes only; bench-side use of the oracle module)
from visrag_amd.evisrag import GenConfig, LLM, SamplingParams  # noqa: E402
from visrag_amd.synth import synth_tensor  # noqa: E402

n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n_new = int(sys.argv[2]) if len(sys.argv) > 2 else 128
n_q = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cfg = GenConfig()
ocfg = QwenGenConfig()
t0 = time.time()
llm = LLM(cfg, limit_mm_per_prompt={"image": 5}, max_model_len=4096, max_prefill=2048)


def weights():
    for k, (shape, amp, off) in weight_specs(ocfg).items():
        yield k, synth_tensor(k, shape, amp, 0, off, device="cuda").to(torch.bfloat16)


llm.load_weights(weights())
torch.cuda.synchronize()
t_load = time.time() - t0
params = sum(int(np.prod(s)) for s, _, _ in weight_specs(ocfg).values())
stream_params = params - cfg.vocab_size * cfg.hidden_size            # the embedding table is gathered, not streamed
# prompt: ~120 text tokens around n_img pages of 16 x 16 merged tokens (448 x 448 page -> 32 x 32 patches -> 2 x 2 merge)
rng = np.random.default_rng(0)
grid = (16, 16)
ids = list(rng.integers(1000, 50000, 60)) + sum(([cfg.image_token_id] * (grid[0] * grid[1]) + [198] for _ in range(n_img)), []) + \
    list(rng.integers(1000, 50000, 60))
embs = [(rng.standard_normal((grid[0] * grid[1], cfg.hidden_size)) * 0.05).astype(np.float32) for _ in range(n_img)]
prompt = {"prompt_token_ids": [int(t) for t in ids], "multi_modal_data": {"image_embeds": embs, "image_grids": [grid] * n_img}}
sp = SamplingParams(temperature=0.1, repetition_penalty=1.05, max_tokens=n_new, stop_token_ids=())
llm.generate([prompt], SamplingParams(temperature=0.1, repetition_penalty=1.05, max_tokens=4, stop_token_ids=()))   # warm-up
pre, dec, tot = [], [], []
for q in range(n_q):
    torch.cuda.synchronize(); a = time.perf_counter()
    pos3 = llm.prefill(prompt["prompt_token_ids"], embs, [grid] * n_img)
    tok = llm.sample(sp, 0)
    torch.cuda.synchronize(); b = time.perf_counter()
    nxt = int(pos3.max()) + 1
    for step in range(1, n_new):
        llm.decode(tok, nxt); nxt += 1
        tok = llm.sample(sp, step)
    torch.cuda.synchronize(); c = time.perf_counter()
    pre.append(b - a); dec.append((c - b) / (n_new - 1)); tot.append(c - a)
T = len(ids)
ms_tok = float(np.median(dec)) * 1e3
print(json.dumps({
    "workload": f"Qwen2.5-VL-7B-shaped language model, bf16, random weights; prompt {T} tokens ({n_img} pages x 256 image tokens as "
                f"embedding rows + text), {n_new} answer tokens, temperature 0.1, repetition_penalty 1.05, one query at a time",
    "params_billion": round(params / 1e9, 3), "load_s": round(t_load, 1),
    "prefill_ms": round(float(np.median(pre)) * 1e3, 2), "prefill_tokens_per_s": round(T / float(np.median(pre))),
    "prefill_tflops": round(2.0 * stream_params * T / float(np.median(pre)) / 1e12, 1),
    "decode_ms_per_token": round(ms_tok, 3), "decode_tokens_per_s": round(1e3 / ms_tok, 1),
    "queries_per_s": round(1.0 / float(np.median(tot)), 3),
    "queries_per_s_at_2048_tokens": round(1.0 / (float(np.median(pre)) + 2047 * float(np.median(dec))), 4),
    "roofline": {"bound": "hbm", "kernel": "decode step (gemm256w_bf16_kernel as a weight streamer, M = 1)",
                 "achieved": round(stream_params * 2 / (ms_tok * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                 "frac": round(stream_params * 2 / (ms_tok * 1e-3) / 8e12, 4),
                 "bytes_per_token": stream_params * 2}}))
