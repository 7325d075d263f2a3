"""Encode-only workload for rocprofv3: full-dims VisRAG-Ret, batch 32 pages x N steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from PIL import Image
from visrag_amd.config import full_config
from visrag_amd.engine import HipEncoder
from visrag_amd.preprocess import prepare_batch
from visrag_amd.synth import iter_synth_weights, synth_pages
from visrag_amd.tokenizer import StandInTokenizer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = full_config(); B = 32
enc = HipEncoder(cfg, max_images=B, max_tokens=4096, max_seqs=64)
enc.load_state_dict(iter_synth_weights(cfg, 0, device="cuda"))
tok = StandInTokenizer(cfg.vocab_size)
pages = synth_pages(B, size=448, seed=0)
items = prepare_batch([""] * B, [Image.fromarray(p) for p in pages], tok, cfg, 2048)
dev = [torch.from_numpy(p).cuda() for p in pages]
out = torch.empty((B, cfg.hidden_size), dtype=torch.float32, device="cuda")
for _ in range(steps):
    enc.encode_items(items, device_slices=dev, out=out)
torch.cuda.synchronize()
print("done", float(out.norm()))
