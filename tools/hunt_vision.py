"""Bug hunt: N random image-grid lists (1..4 images, 2..24 patches a side, even) through the generator's vision tower (both
fixture tower shapes: head_dim 40 and 80; windowed + full-attention blocks) against oracle/qwen_vision_oracle.py.
    python tools/hunt_vision.py 60        (round 3: 120 cases, worst error 6e-3 of scale, 0 failures)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle.qwen_gen_oracle import synth_weights, tiny_config
from oracle.qwen_vision_oracle import QwenVisionOracle, hd80_vision_config, synth_vision_weights, tiny_vision_config
from visrag_amd.evisrag import GenConfig, LLM, VisionConfig

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cfg = tiny_config()
gc = GenConfig(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
               num_key_value_heads=cfg.num_key_value_heads, intermediate_size=cfg.intermediate_size, vocab_size=cfg.vocab_size,
               rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta, mrope_section=tuple(cfg.mrope_section), image_token_id=5, eos_token_ids=())
bad, worst_rel, worst_cos = [], 0.0, 1.0
for vname, vcfg in (("hd40", tiny_vision_config(256)), ("hd80", hd80_vision_config(256))):
    wv = synth_vision_weights(vcfg, seed=3)
    vc = VisionConfig(depth=vcfg.depth, hidden_size=vcfg.hidden_size, num_heads=vcfg.num_heads, intermediate_size=vcfg.intermediate_size,
                      out_hidden_size=vcfg.out_hidden_size, window_size=vcfg.window_size, fullatt_block_indexes=tuple(vcfg.fullatt_block_indexes))
    w = dict(synth_weights(cfg, seed=7)); w.update(wv)
    llm = LLM(gc, max_model_len=2048, max_prefill=1024, vision=vc, max_vision_rows=4096, weights=w)
    o = QwenVisionOracle(vcfg, wv)
    pdim = 3 * 2 * 14 * 14
    for seed in range(n):
        rng = np.random.default_rng(seed)
        grids = [(1, 2 * int(rng.integers(1, 13)), 2 * int(rng.integers(1, 13))) for _ in range(int(rng.integers(1, 5)))]
        rows = sum(t * h * w_ for t, h, w_ in grids)
        if rows > 4096:
            continue
        px = torch.from_numpy(rng.standard_normal((rows, pdim)).astype(np.float32)).to(torch.bfloat16).float()
        try:
            ours = llm.encode_images(px.numpy(), np.asarray(grids, dtype=np.int32))
            ref = o.forward(px, grids).numpy()
            rel = float(np.abs(ours - ref).max() / np.abs(ref).max())
            cos = float(((ours * ref).sum(-1) / (np.linalg.norm(ours, axis=-1) * np.linalg.norm(ref, axis=-1))).min())
            worst_rel, worst_cos = max(worst_rel, rel), min(worst_cos, cos)
            if rel > 2e-2 or cos < 1 - 1e-3:
                bad.append((vname, seed, grids, rel, cos))
        except Exception as ex:
            bad.append((vname, seed, grids, repr(ex)[:200]))
    llm.close()
print("cases", 2 * n, "failures", len(bad), "worst error / scale", worst_rel, "worst cosine", worst_cos)
for b in bad[:10]: print(b)
