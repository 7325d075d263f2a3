"""Experiment: do two encode batches in flight (two HipEncoder instances on two HIP streams) beat one?
(tails of one batch's kernels filled by the other's workgroups, LayerNorms under the other's GEMMs)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from PIL import Image
from visrag_amd.config import full_config
from visrag_amd.engine import HipEncoder, overlapping_streams
from visrag_amd.preprocess import prepare_batch
from visrag_amd.synth import iter_synth_weights, synth_pages
from visrag_amd.tokenizer import StandInTokenizer
cfg = full_config(); B = 32
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
e0 = HipEncoder(cfg, max_images=B, max_tokens=4096, max_seqs=64)
e0.load_state_dict(iter_synth_weights(cfg, 0, device="cuda"))
encs = [e0] + [e0.clone() for _ in range(N - 1)]
tok = StandInTokenizer(cfg.vocab_size)
pages = synth_pages(B, size=448, seed=0)
items = prepare_batch([""] * B, [Image.fromarray(p) for p in pages], tok, cfg, 2048)
dev = [torch.from_numpy(p).cuda() for p in pages]
outs = [torch.empty((B, cfg.hidden_size), dtype=torch.float32, device="cuda") for _ in range(N)]
streams = overlapping_streams(0, N)       # (pool streams can share a hardware queue: probed)
def run(n, two):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        j = i % N if two else 0
        with torch.cuda.stream(streams[j]):
            encs[j].encode_items(items, device_slices=dev, out=outs[j])
    torch.cuda.synchronize()
    return B * n / (time.perf_counter() - t0)
run(2 * N, True)
for rep in range(2):
    print("one stream  %.1f pages/s" % run(24, False))
    print("%d streams %.1f pages/s" % (N, run(24, True)))
ref = outs[0].clone(); 
print("outputs equal across instances:", bool(torch.equal(outs[0], outs[1])))
