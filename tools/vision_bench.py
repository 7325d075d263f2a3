"""EVisRAG vision tower alone at Qwen2.5-VL-7B shape (random weights, a one-layer language model behind it so that the
model finalizes quickly):
    python tools/vision_bench.py [n_images=5] [side_patches=32] [iters=5]   -> one JSON line
Times vg_vision_encode on device-resident work only as far as the ABI allows: the pixel rows come from the host
(24 MB for five 448 x 448 pages), so the line reports the call and, separately, the same call minus a measured H2D
copy of that size.  Profile with: rocprofv3 --kernel-trace --stats -- python tools/vision_bench.py"""
import itertools
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visrag_amd.evisrag import (LLM, GenConfig, VisionConfig, iter_synth_gen_weights, iter_synth_vision_weights,  # noqa: E402
                                vision_weight_specs)

a = [int(x) for x in sys.argv[1:]]
n_images, side, iters = (a + [5, 32, 5][len(a):])
cfg = GenConfig(num_hidden_layers=1, intermediate_size=512, vocab_size=1024)
vc = VisionConfig()
rows = n_images * side * side
llm = LLM(cfg, limit_mm_per_prompt={"image": n_images}, max_model_len=max(4096, rows // 4 + 64), max_prefill=max(2048, rows // 4 + 64),
          vision=vc, max_vision_rows=rows)
llm.load_weights(itertools.chain(iter_synth_gen_weights(cfg, 0, device="cuda:0", bf16=True),
                                 iter_synth_vision_weights(vc, 0, device="cuda:0", bf16=True)))
rng = np.random.default_rng(0)
px = rng.standard_normal((rows, vc.patch_dim)).astype(np.float32)
thw = np.asarray([(1, side, side)] * n_images, dtype=np.int32)
llm.encode_images(px, thw, fetch=False)
ts = []
for _ in range(iters):
    torch.cuda.synchronize(); t = time.perf_counter()
    llm.encode_images(px, thw, fetch=False)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
dst = torch.empty(px.size, dtype=torch.float32, device="cuda:0")
src = torch.from_numpy(px.reshape(-1))
cs = []
for _ in range(iters):
    torch.cuda.synchronize(); t = time.perf_counter()
    dst.copy_(src)
    torch.cuda.synchronize(); cs.append(time.perf_counter() - t)
specs = vision_weight_specs(vc)
blk = sum(int(np.prod(sh)) for k, (sh, _, _) in specs.items() if ".blocks." in k and k.endswith("weight") and len(sh) == 2)
mrg = sum(int(np.prod(sh)) for k, (sh, _, _) in specs.items() if ".merger.mlp." in k and k.endswith("weight"))
win = 64
att = sum(4.0 * rows * (side * side if l in vc.fullatt_block_indexes else win) * vc.hidden_size for l in range(vc.depth))
flop = 2.0 * rows * (blk + vc.hidden_size * vc.patch_dim) + 2.0 * (rows / 4) * mrg + att
ms, h2d = float(np.median(ts)) * 1e3, float(np.median(cs)) * 1e3
print(json.dumps({"workload": f"{n_images} pages of {side} x {side} patches ({rows} patch rows -> {rows // 4} image tokens), Qwen2.5-VL-7B tower, bf16, random weights",
                  "call_ms": round(ms, 2), "h2d_ms": round(h2d, 2), "device_ms": round(ms - h2d, 2), "algorithmic_tflop": round(flop / 1e12, 3),
                  "tflops_call": round(flop / ms / 1e9, 1), "tflops_device": round(flop / (ms - h2d) / 1e9, 1),
                  "pages_per_s": round(n_images / ms * 1e3, 1)}))
llm.close()
