import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from PIL import Image
from oracle import visrag_ret_oracle as O
from visrag_amd.config import full_config
from visrag_amd.synth import iter_synth_weights, synth_pages
from visrag_amd.preprocess import prepare_batch
from visrag_amd.tokenizer import StandInTokenizer
cfg = full_config()
W = {k: v.cpu() for k, v in iter_synth_weights(cfg, 0, device="cuda")}
tok = StandInTokenizer(cfg.vocab_size)
it = prepare_batch([""], [Image.fromarray(synth_pages(1)[0])], tok, cfg)
for th in (16, 32, 64, 128):
    torch.set_num_threads(th)
    t = time.perf_counter()
    O.encode(W, cfg, [it[0].input_ids], [it[0].image_bound], [it[0].slices])
    print(th, "threads:", round(time.perf_counter() - t, 2), "s/page", flush=True)
