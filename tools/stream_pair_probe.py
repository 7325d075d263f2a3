#!/usr/bin/env python
"""Two batches in flight on two torch streams: does the rate depend on WHICH two streams of torch's pool?  (HIP maps streams onto a
few hardware queues; two streams on one queue run one behind the other.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from PIL import Image
from visrag_amd.config import full_config
from visrag_amd.engine import HipEncoder, overlapping_streams, streams_overlap
from visrag_amd.preprocess import prepare_batch
from visrag_amd.synth import iter_synth_weights, synth_pages
from visrag_amd.tokenizer import StandInTokenizer

B = 32
cfg = full_config()
dev = torch.device("cuda:0")
enc = HipEncoder(cfg, device=0, max_images=B, max_tokens=4096, max_seqs=64)
enc.load_state_dict(iter_synth_weights(cfg, 0, device=dev))
enc2 = enc.clone()
tok = StandInTokenizer(cfg.vocab_size)
pages = synth_pages(B, size=448, seed=0)
items = prepare_batch([""] * B, [Image.fromarray(p) for p in pages], tok, cfg, 2048)
px = [torch.from_numpy(p).to(dev) for p in pages]
outs = [torch.empty((B, cfg.hidden_size), device=dev) for _ in range(2)]
streams = [torch.cuda.Stream(device=dev) for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12)]
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"), " stream handles:", [hex(s.cuda_stream) for s in streams[:6]])

def rate(sa, sb, n=16):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n):
            with torch.cuda.stream(sb if i & 1 else sa):
                (enc2 if i & 1 else enc).encode_items(items, device_slices=px, out=outs[i & 1])
        torch.cuda.synchronize()
    return n * B / (time.perf_counter() - t0)

for i in range(len(streams) - 1):
    print(f"streams {i},{i + 1}: {rate(streams[i], streams[i + 1]):6.1f} pages/s   vr_streams_overlap: {streams_overlap(streams[i], streams[i + 1])}", flush=True)
for i, j in ((0, 2), (0, 4), (1, 5), (2, 6), (0, 3)):
    print(f"streams {i},{j}: {rate(streams[i], streams[j]):6.1f} pages/s", flush=True)
print(f"default + stream 0: {rate(torch.cuda.default_stream(dev), streams[0]):6.1f} pages/s")
sa, sb = overlapping_streams(dev, 2)
print(f"overlapping_streams(): {rate(sa, sb):6.1f} pages/s")
