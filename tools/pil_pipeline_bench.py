#!/usr/bin/env python
"""PIL pages -> distributed_parallel_embedding_inference -> pickle shard (the reference's own entry point), pages/s, and
the host-only share of a batch (prepare without encode):   python tools/pil_pipeline_bench.py [pages=256] [batch=32]"""
import os, sys, tempfile, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from PIL import Image
from visrag_amd.config import full_config
from visrag_amd.gpu_resize import prepare_item_gpu
from visrag_amd.inference import distributed_parallel_embedding_inference
from visrag_amd.modeling import DRModelForInference
from visrag_amd.synth import iter_synth_weights, synth_pages
from visrag_amd.tokenizer import StandInTokenizer

n, B = (int(sys.argv[1]) if len(sys.argv) > 1 else 256), (int(sys.argv[2]) if len(sys.argv) > 2 else 32)
cfg = full_config()
model = DRModelForInference.build(cfg=cfg, state_dict=iter_synth_weights(cfg, 0, device="cuda"), max_images=B, max_tokens=4096,
                                  max_seqs=64, pipeline=2)
tok = StandInTokenizer(cfg.vocab_size)
pages = synth_pages(64, size=448, seed=0)
corpus = [{"id": str(i), "text": "", "image": Image.fromarray(pages[i % 64])} for i in range(n)]
extra = {"tokenizer": tok, "max_inp_length": 2048}
with tempfile.TemporaryDirectory() as td:
    a = types.SimpleNamespace(output_dir=td, per_device_eval_batch_size=B, process_index=0, world_size=1, max_inmem_docs=10_000_000,
                              device="cuda:0")
    distributed_parallel_embedding_inference(corpus[:2 * B], model, a, "corpus", False, extra)
    host = {"t": 0.0, "n": 0}
    fwd = model.forward

    def timed_forward(*aa, **kk):          # time the calling thread spends inside model(...): host prepare + launches
        t = time.perf_counter()
        r = fwd(*aa, **kk)
        host["t"] += time.perf_counter() - t; host["n"] += 1
        return r
    model.forward = timed_forward
    model.__class__.__call__ = lambda self, *aa, **kk: self.forward(*aa, **kk)
    for workers in (0, 1, 4, 1):
        a.dataloader_num_workers = workers
        host["t"], host["n"] = 0.0, 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        distributed_parallel_embedding_inference(corpus, model, a, "corpus", False, extra)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"pil_pipeline workers={workers}: {n / dt:.1f} pages/s ({dt / (n / B) * 1e3:.1f} ms per batch of {B}; "
              f"{host['t'] / max(host['n'], 1) * 1e3:.1f} ms of it inside model(...) on the calling thread)", flush=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for lo in range(0, n, B):
    [prepare_item_gpu("", c["image"], tok, cfg, 2048, 0) for c in corpus[lo:lo + B]]
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"host prepare + upload alone: {dt / (n / B) * 1e3:.1f} ms per batch of {B}")
