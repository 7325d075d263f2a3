"""Bug hunt for the split-precision text pass on its two GEMM routes (weight streamer for <= 32 rows, hi|lo-stacked tile GEMM
beyond): random text-only batches on a decoder whose widths are multiples of 256 (the streamer's tile), weights that are NOT
bf16-representable (all three operand-half products run), against the CPU oracle.      python tools/hunt_text_stream.py 60"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import visrag_ret_oracle as O
from visrag_amd.config import tiny_config
from visrag_amd.engine import HipEncoder
from visrag_amd.modeling import DRModelForInference
from visrag_amd.preprocess import prepare_batch
from visrag_amd.synth import synth_queries, synth_state_dict
from visrag_amd.tokenizer import StandInTokenizer

cfg = tiny_config()
cfg.hidden_size, cfg.num_heads, cfg.intermediate_size, cfg.num_layers = 512, 8, 1280, 3
W = synth_state_dict(cfg, 0)
for k_ in list(W.keys()):
    if k_.startswith("llm.model.layers.") and k_.endswith("proj.weight"):
        W[k_] = (W[k_] * 1.0009765625).contiguous()
enc = HipEncoder(cfg, max_images=2, max_tokens=2048, max_seqs=16)
enc.load_state_dict(((k_, v.cuda()) for k_, v in W.items()))
model = DRModelForInference(cfg, enc, gpu_preprocess=False)
tok = StandInTokenizer(cfg.vocab_size)
n, bad, worst, routes = int(sys.argv[1]) if len(sys.argv) > 1 else 60, [], 0.0, {"stream": 0, "tiles": 0}
for seed in range(n):
    rng = np.random.default_rng(seed)
    B = int(rng.integers(1, 9))
    texts = [synth_queries(1, seed=seed * 10 + i, min_words=1, max_words=int(rng.choice([2, 6, 14, 30, 120])))[0] for i in range(B)]
    items = prepare_batch(texts, [None] * B, tok, cfg, int(rng.choice([8, 16, 64, 512])))
    T = sum(len(it.input_ids) for it in items)
    routes["stream" if T <= 32 else "tiles"] += 1
    try:
        ref = O.encode(W, cfg, [it.input_ids for it in items], [[]] * B, [[]] * B).numpy()
        got = model.encode_prepared(items).cpu().numpy()
        err = float(1 - (got * ref).sum(1).min())
        worst = max(worst, err)
        if not (err < 2e-6) or not np.isfinite(got).all():
            bad.append((seed, err, [len(it.input_ids) for it in items]))
    except Exception as e:
        bad.append((seed, repr(e)[:200]))
print("configs", n, routes, "failures", len(bad), "worst 1 - cosine", worst)
for b in bad[:10]: print(b)
