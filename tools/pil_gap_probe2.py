#!/usr/bin/env python
"""bench.py's PIL leg measures 0.94 of its own two-stream loop, tools/pil_gap_probe.py 0.99 in a fresh process: which piece of
bench.py's history does it?  The PIL leg after each of them."""
import os, sys, tempfile, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from PIL import Image
from visrag_amd.config import full_config
from visrag_amd.engine import HipEncoder, HipIndex
from visrag_amd.inference import distributed_parallel_embedding_inference
from visrag_amd.modeling import DRModelForInference
from visrag_amd.preprocess import prepare_batch
from visrag_amd.synth import iter_synth_weights, synth_pages
from visrag_amd.tokenizer import StandInTokenizer

B, NB = 32, 48
cfg = full_config()
dev = torch.device("cuda:0")
enc = HipEncoder(cfg, device=0, max_images=B, max_tokens=4096, max_seqs=64)
enc.load_state_dict(iter_synth_weights(cfg, 0, device=dev))
tok = StandInTokenizer(cfg.vocab_size)
pages = synth_pages(64, size=448, seed=0)
pil = [Image.fromarray(p) for p in pages]
corpus = [{"id": str(i), "text": "", "image": pil[i % 64]} for i in range(NB * B)]
extra = {"tokenizer": tok, "max_inp_length": 2048}
td = tempfile.mkdtemp()
a = types.SimpleNamespace(output_dir=td, per_device_eval_batch_size=B, process_index=0, world_size=1, max_inmem_docs=1024,
                          device="cuda:0", dataloader_num_workers=1)

def pil_leg(tag):
    model = DRModelForInference(cfg, enc)
    model.set_pipeline(2)
    distributed_parallel_embedding_inference(corpus[:2 * B], model, a, "corpus", False, extra)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    distributed_parallel_embedding_inference(corpus, model, a, "corpus", True, extra)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{tag:44s}: {NB * B / dt:7.1f} pages/s", flush=True)
    for e, _ in model._slots[1:]:
        e.close()

pil_leg("fresh process")
items = prepare_batch([""] * 64, pil, tok, cfg, 2048)
dev_pages = [torch.from_numpy(p).to(dev) for p in pages]
batches = [(items[i:i + B], dev_pages[i:i + B]) for i in range(0, 64, B)]
index = HipIndex(cfg.hidden_size, 100_000 + 1024, device=0)
out = torch.empty((B, cfg.hidden_size), dtype=torch.float32, device=dev)
for i in range(12):
    it, px = batches[i % 2]
    enc.encode_items(it, device_slices=px, out=out)        # (torch's default stream, like bench.py's step loop)
    index.add(out)
torch.cuda.synchronize()
pil_leg("after the step loop on the default stream")
for lvl in (True, 2):
    enc.set_profile(lvl)
    for i in range(6):
        it, px = batches[i % 2]
        enc.encode_items(it, device_slices=px, out=out)
    torch.cuda.synchronize()
    enc.get_profile()
enc.set_profile(False)
pil_leg("after the profile passes")
enc2 = enc.clone()
slots = [(enc, torch.cuda.Stream(device=dev), out), (enc2, torch.cuda.Stream(device=dev), torch.empty_like(out))]
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(24):
    e, st, o = slots[i & 1]
    it, px = batches[i % 2]
    with torch.cuda.stream(st):
        e.encode_items(it, device_slices=px, out=o)
torch.cuda.synchronize()
print(f"{'bare two-stream loop':44s}: {24 * B / (time.perf_counter() - t0):7.1f} pages/s")
enc2.close()
pil_leg("after the two-stream loop + close")
