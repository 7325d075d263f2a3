"""Bug hunt: N random (prompt length, image grids, decode steps) cases of the generator's language model at the fixtures' tiny
shape against oracle/qwen_gen_oracle.py: prefill logits, then teacher-forced decode steps (each compared with the oracle's
incremental forward).    python tools/hunt_generate.py 150        (round 3: 150 cases, worst error 1.2e-2 of the logits' scale, 0 failures)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle.qwen_gen_oracle import QwenGenOracle, synth_weights, tiny_config
from visrag_amd.evisrag import GenConfig, LLM, rope_index

cfg = tiny_config()
g = GenConfig(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
              num_key_value_heads=cfg.num_key_value_heads, intermediate_size=cfg.intermediate_size, vocab_size=cfg.vocab_size,
              rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta, mrope_section=tuple(cfg.mrope_section), image_token_id=5, eos_token_ids=())
w = synth_weights(cfg, seed=7)
llm = LLM(g, max_model_len=1024, max_prefill=768, weights=w)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad, worst = [], 0.0
for seed in range(n):
    rng = np.random.default_rng(seed)
    n_img = int(rng.integers(0, 3))
    grids = [(int(rng.integers(1, 7)) * 1, int(rng.integers(1, 7)) * 1) for _ in range(n_img)]    # image tokens per page: h x w (after the merger)
    ids = []
    for gh, gw in grids:
        ids += rng.integers(16, cfg.vocab_size, int(rng.integers(0, 40))).tolist() + [5] * (gh * gw)
    ids += rng.integers(16, cfg.vocab_size, int(rng.integers(1, 400))).tolist()
    ids = ids[:700]
    embs = [(0.05 * rng.standard_normal((gh * gw, cfg.hidden_size))).astype(np.float32) for gh, gw in grids]
    steps = int(rng.integers(0, 6))
    toks = rng.integers(16, cfg.vocab_size, steps).tolist()
    try:
        if sum(1 for t in ids if t == 5) != sum(gh * gw for gh, gw in grids):
            continue
        pos3 = llm.prefill(ids, embs, grids)
        o = QwenGenOracle(cfg, w)                  # (the oracle keeps a KV cache: a fresh one per sequence)
        idt = torch.tensor(ids)
        emb = o.embed(idt).clone()
        if embs:
            emb[idt == 5] = torch.from_numpy(np.concatenate(embs))
        pos = torch.from_numpy(pos3).long()
        ref = o.forward(emb, pos)[-1].numpy()
        ours = llm.logits()
        e = float(np.abs(ours - ref).max() / np.abs(ref).max()); worst = max(worst, e)
        if e > 1.5e-2:
            bad.append((seed, "prefill", len(ids), grids, e))
        nxt = int(pos3.max()) + 1
        for k, t in enumerate(toks):
            llm.decode(t, nxt + k)
            ref = o.forward(o.embed(torch.tensor([t])), torch.full((3, 1), nxt + k))[-1].numpy()   # appended to its cache
            ours = llm.logits()
            e = float(np.abs(ours - ref).max() / np.abs(ref).max()); worst = max(worst, e)
            if e > 2e-2:
                bad.append((seed, "decode", k, len(ids), grids, e))
    except Exception as ex:
        bad.append((seed, repr(ex)[:200]))
print("cases", n, "failures", len(bad), "worst error / scale", worst)
for b in bad[:10]: print(b)
