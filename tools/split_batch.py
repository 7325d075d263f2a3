"""Experiment: ONE batch of 32 pages encoded as two half batches on two HIP streams (two workspaces over the same
weights) against the whole batch on one stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from PIL import Image
from visrag_amd.config import full_config
from visrag_amd.engine import HipEncoder
from visrag_amd.preprocess import prepare_batch
from visrag_amd.synth import iter_synth_weights, synth_pages
from visrag_amd.tokenizer import StandInTokenizer
cfg = full_config(); B = 32
e0 = HipEncoder(cfg, max_images=B, max_tokens=4096, max_seqs=64)
e0.load_state_dict(iter_synth_weights(cfg, 0, device="cuda"))
e1 = e0.clone()
tok = StandInTokenizer(cfg.vocab_size)
pages = synth_pages(B, size=448, seed=0)
items = prepare_batch([""] * B, [Image.fromarray(p) for p in pages], tok, cfg, 2048)
dev = [torch.from_numpy(p).cuda() for p in pages]
out = torch.empty((B, cfg.hidden_size), dtype=torch.float32, device="cuda")
ref = torch.empty_like(out)
side = torch.cuda.Stream()
def whole(o):
    e0.encode_items(items, device_slices=dev, out=o)
def halves(o, cut):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        e1.encode_items(items[cut:], device_slices=dev[cut:], out=o[cut:])
    e0.encode_items(items[:cut], device_slices=dev[:cut], out=o[:cut])
    main.wait_stream(side)
def run(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return B * n / (time.perf_counter() - t0)
whole(ref)
for rep in range(2):
    print("whole batch, one stream   %.1f pages/s" % run(lambda: whole(out)))
    for cut in (16, 12, 8):
        print("split %2d/%2d, two streams   %.1f pages/s" % (cut, B - cut, run(lambda: halves(out, cut))))
halves(out, 16); torch.cuda.synchronize()
print("max |split - whole| =", float((out - ref).abs().max()))
