"""In-model A/B of a kernel build: encode `steps` batches of 32 synthetic pages (full dims) with the
library named by VISRAG_HIP_LIB (default: the product build), print one JSON line with ms/step and the
per-kernel-class HIP-event times.  Several rounds per process (interleave processes for A/B):

    VISRAG_HIP_LIB=visrag_amd/libvisrag_hip_p1.so python tools/ab_encode.py 10 3
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from PIL import Image
from visrag_amd.config import full_config
from visrag_amd.engine import HipEncoder
from visrag_amd.preprocess import prepare_batch
from visrag_amd.synth import iter_synth_weights, synth_pages
from visrag_amd.tokenizer import StandInTokenizer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = full_config(); B = 32
enc = HipEncoder(cfg, max_images=B, max_tokens=4096, max_seqs=64)
enc.load_state_dict(iter_synth_weights(cfg, 0, device="cuda"))
tok = StandInTokenizer(cfg.vocab_size)
pages = synth_pages(B, size=448, seed=0)
items = prepare_batch([""] * B, [Image.fromarray(p) for p in pages], tok, cfg, 2048)
dev = [torch.from_numpy(p).cuda() for p in pages]
out = torch.empty((B, cfg.hidden_size), dtype=torch.float32, device="cuda")
for _ in range(3):
    enc.encode_items(items, device_slices=dev, out=out)
torch.cuda.synchronize()
res = []
for r in range(rounds):
    enc.set_profile(int(os.environ.get("AB_PROFILE", "1")))     # 2: the decoder's sub-phases
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        enc.encode_items(items, device_slices=dev, out=out)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    prof = enc.get_profile(); enc.set_profile(False)
    res.append({"ms_per_step": round(dt / steps * 1e3, 3), "pages_per_s": round(B * steps / dt, 1),
                "phases_ms": {k: round(v["ms"] / steps, 3) for k, v in prof.items()},
                "phases_tf": {k: round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 1) for k, v in prof.items()}})
best = min(res, key=lambda x: x["ms_per_step"])
print(json.dumps({"lib": os.environ.get("VISRAG_HIP_LIB", "product"), "best": best,
                  "all_ms": [x["ms_per_step"] for x in res], "checksum": float(out.double().abs().sum())}))
