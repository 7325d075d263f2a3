#!/usr/bin/env python
"""Where do the 6 % between `pil_pipeline` and `pipelined` go?  The reference's entry point over 64 batches of 32 pages handed in
as (a) PIL images, (b) u8 numpy arrays, (c) u8 cuda tensors (no upload), two batches in flight, next to the bare two-stream loop.
    python tools/pil_gap_probe.py"""
import os, sys, tempfile, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image
from visrag_amd.config import full_config
from visrag_amd.inference import distributed_parallel_embedding_inference
from visrag_amd.modeling import DRModelForInference
from visrag_amd.synth import iter_synth_weights, synth_pages
from visrag_amd.tokenizer import StandInTokenizer

B, NB = 32, 64
cfg = full_config()
model = DRModelForInference.build(cfg=cfg, state_dict=iter_synth_weights(cfg, 0, device="cuda"), max_images=B, max_tokens=4096,
                                  max_seqs=64, pipeline=2)
tok = StandInTokenizer(cfg.vocab_size)
pages = synth_pages(64, size=448, seed=0)
kinds = {"pil": [Image.fromarray(p) for p in pages], "numpy": [np.ascontiguousarray(p) for p in pages],
         "cuda": [torch.from_numpy(p).cuda() for p in pages]}
extra = {"tokenizer": tok, "max_inp_length": 2048}
with tempfile.TemporaryDirectory() as td:
    a = types.SimpleNamespace(output_dir=td, per_device_eval_batch_size=B, process_index=0, world_size=1, max_inmem_docs=1024,
                              device="cuda:0", dataloader_num_workers=1)
    for kind, imgs in kinds.items():
        corpus = [{"id": str(i), "text": "", "image": imgs[i % 64]} for i in range(NB * B)]
        distributed_parallel_embedding_inference(corpus[:2 * B], model, a, "corpus", False, extra)
        for workers in (1, 0):
            a.dataloader_num_workers = workers
            torch.cuda.synchronize(); t0 = time.perf_counter()
            distributed_parallel_embedding_inference(corpus, model, a, "corpus", True, extra)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print(f"{kind:6s} workers={workers}: {NB * B / dt:7.1f} pages/s  ({dt / NB * 1e3:.2f} ms per batch)", flush=True)
# the bare loop: prepared items, device pixels, two encoders on two streams
from visrag_amd.preprocess import prepare_batch
items = prepare_batch([""] * B, kinds["pil"][:B], tok, cfg, 2048)
dev = kinds["cuda"][:B]
slots = [(e, s or torch.cuda.Stream()) for e, s in model._slots]
outs = [torch.empty((B, cfg.hidden_size), device="cuda") for _ in slots]
def step(i):
    e, st = slots[i & 1]
    with torch.cuda.stream(st):
        e.encode_items(items, device_slices=dev, out=outs[i & 1])
for i in range(4): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(NB): step(i)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"bare two-stream loop: {NB * B / dt:7.1f} pages/s  ({dt / NB * 1e3:.2f} ms per batch)")
