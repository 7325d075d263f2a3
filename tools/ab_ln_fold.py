#!/usr/bin/env python
"""In-model A/B on one box: the encode step with the ViT blocks' LayerNorms as launches (default) against the same step with
them folded into the GEMMs around them (VR_VIT_LN_FOLD=1 at vr_model_create; experimental, round 4 -> 5).
    python tools/ab_ln_fold.py [steps=12]
Prints ms/step (min / median over alternating rounds), the phase table of each, and the cosine between the two outputs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image
from visrag_amd.config import full_config
from visrag_amd.engine import HipEncoder
from visrag_amd.preprocess import prepare_batch
from visrag_amd.synth import iter_synth_weights, synth_pages
from visrag_amd.tokenizer import StandInTokenizer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
quick = len(sys.argv) > 2          # only the default and knob 2, two rounds
cfg, B = full_config(), 32
tok = StandInTokenizer(cfg.vocab_size)
pages = synth_pages(B, size=448, seed=0)
items = prepare_batch([""] * B, [Image.fromarray(p) for p in pages], tok, cfg, 2048)
dev = [torch.from_numpy(p).cuda() for p in pages]
encs = {}
for name, knob in ((("layernorm launches", None), ("folded, in-kernel stats", "2")) if quick else
                   (("layernorm launches", None), ("folded, stats launch", "1"), ("folded, in-kernel stats", "2"))):
    if knob:
        os.environ["VR_VIT_LN_FOLD"] = knob
    try:
        e = HipEncoder(cfg, max_images=B, max_tokens=4096, max_seqs=64)
    finally:
        os.environ.pop("VR_VIT_LN_FOLD", None)
    e.load_state_dict(iter_synth_weights(cfg, 0, device="cuda"))
    encs[name] = e
outs = {n: torch.empty((B, cfg.hidden_size), device="cuda") for n in encs}

def run(name, n):
    e = encs[name]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        e.encode_items(items, device_slices=dev, out=outs[name])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for n in encs: run(n, 3)
ms = {n: [] for n in encs}
for rnd in range(2 if quick else 4):
    for n in encs:
        ms[n].append(run(n, steps))
for n in encs:
    e = encs[n]
    e.set_profile(True); run(n, steps); prof = e.get_profile(); e.set_profile(False)
    print(f"{n:20s} ms/step min {min(ms[n]):.3f} med {sorted(ms[n])[len(ms[n]) // 2]:.3f}   " +
          "  ".join(f"{k} {v['ms'] / steps:.3f}" for k, v in prof.items() if not k.startswith('dec_')))
base = outs["layernorm launches"].cpu().numpy()
for n in list(encs)[1:]:
    b = outs[n].cpu().numpy()
    print(f"{n}: cosine against the default outputs: min {float((base * b).sum(1).min()):.7f}  finite: {bool(np.isfinite(b).all())}")
