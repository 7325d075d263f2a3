"""Soak of the opt-in persistent decode kernel (VR_DECODE_PERSIST=1): N free-running tokens at the 7B shape (4 layers), the token
stream compared with the separate launches' on the same seeds; reports ms per token of both."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle.qwen_gen_oracle import QwenGenConfig, synth_weights
from visrag_amd.evisrag import GenConfig, LLM, SamplingParams

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
cfg = QwenGenConfig(num_hidden_layers=4)
g = GenConfig(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
              num_key_value_heads=cfg.num_key_value_heads, intermediate_size=cfg.intermediate_size, vocab_size=cfg.vocab_size,
              rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta, mrope_section=tuple(cfg.mrope_section), image_token_id=5, eos_token_ids=())
ids = np.random.default_rng(2).integers(16, cfg.vocab_size, 300).tolist()
sp = SamplingParams(temperature=0.8, repetition_penalty=1.05, max_tokens=n, seed=5)
out = {}
for run, mode in enumerate(("0", "1", "1")):
    os.environ["VR_DECODE_PERSIST"] = mode
    llm = LLM(g, weights=synth_weights(cfg, seed=3, device="cuda"), max_model_len=4096, max_prefill=512)
    llm.prefill(ids)
    llm.sample(sp, 0)
    llm.run_begin(len(ids), sp)
    t0 = time.perf_counter()
    toks = []
    for i in range(n):
        llm.run_step()
        if i >= 4:
            toks.append(llm.run_token(i - 4))
    for i in range(max(n - 4, 0), n):
        toks.append(llm.run_token(i))
    dt = time.perf_counter() - t0
    llm.run_end()
    llm.close()
    out[run] = toks
    print(f"VR_DECODE_PERSIST={mode}: {n} tokens, {dt / n * 1e3:.3f} ms per token (4 layers)")
def first_diff(a, b):
    for i, (x, y) in enumerate(zip(a, b)):
        if x != y:
            return i
    return None
print("first difference, launches vs persistent run 1:", first_diff(out[0], out[1]), " run 2:", first_diff(out[0], out[2]),
      " persistent run 1 vs run 2:", first_diff(out[1], out[2]))
