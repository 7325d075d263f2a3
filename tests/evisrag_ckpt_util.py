"""A tiny Qwen2.5-VL checkpoint DIRECTORY for the tests of `LLM(model=<dir>)`: config.json in the flat layout of
transformers 4.51 (what EVisRAG-7B ships), preprocessor / generation configs, safetensors with that version's key names
("model.layers.*", "visual.*"), and a word-level tokenizer that knows the vision special tokens."""
import json
import os

import numpy as np
import torch

from oracle.qwen_gen_oracle import synth_weights, tiny_config
from oracle.qwen_vision_oracle import tiny_vision_config

GOLD = os.path.join(os.path.dirname(__file__), "golden", "evisrag_vision_tiny.npz")
WORDS = ["<unk>", "<pad>", "<bos>", "<|im_end|>", "<|endoftext|>", "<|image_pad|>", "<|vision_start|>", "<|vision_end|>"] + \
        [f"w{i}" for i in range(8, 1024)]


def vision_weights_from_fixture():
    g = np.load(GOLD)
    return {k[2:]: torch.from_numpy(g[k]).view(torch.bfloat16) for k in g.files if k.startswith("w:")}


def old_key(k: str) -> str:
    if k.startswith("model.visual."):
        return k[len("model."):]
    if k.startswith("model.language_model."):
        return "model." + k[len("model.language_model."):]
    return k


def make_tiny_checkpoint(path: str) -> dict:
    from safetensors.torch import save_file
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    cfg, vcfg = tiny_config(), tiny_vision_config(256)
    os.makedirs(path, exist_ok=True)
    config = {
        "architectures": ["Qwen2_5_VLForConditionalGeneration"], "model_type": "qwen2_5_vl",
        "hidden_size": cfg.hidden_size, "num_hidden_layers": cfg.num_hidden_layers, "num_attention_heads": cfg.num_attention_heads,
        "num_key_value_heads": cfg.num_key_value_heads, "intermediate_size": cfg.intermediate_size, "vocab_size": cfg.vocab_size,
        "rms_norm_eps": cfg.rms_norm_eps, "rope_theta": cfg.rope_theta, "rope_scaling": {"type": "mrope", "mrope_section": list(cfg.mrope_section)},
        "tie_word_embeddings": False, "image_token_id": 5, "vision_start_token_id": 6, "vision_end_token_id": 7, "eos_token_id": 3,
        "vision_config": {"depth": vcfg.depth, "hidden_size": vcfg.hidden_size, "num_heads": vcfg.num_heads,
                          "intermediate_size": vcfg.intermediate_size, "out_hidden_size": vcfg.out_hidden_size, "in_chans": 3,
                          "patch_size": 14, "spatial_merge_size": 2, "temporal_patch_size": 2, "window_size": vcfg.window_size,
                          "fullatt_block_indexes": list(vcfg.fullatt_block_indexes)}}
    json.dump(config, open(os.path.join(path, "config.json"), "w"))
    json.dump({"min_pixels": 56 * 56, "max_pixels": 28 * 28 * 24, "patch_size": 14, "merge_size": 2, "temporal_patch_size": 2,
               "image_mean": [0.48145466, 0.4578275, 0.40821073], "image_std": [0.26862954, 0.26130258, 0.27577711]},
              open(os.path.join(path, "preprocessor_config.json"), "w"))
    json.dump({"eos_token_id": [3, 4], "repetition_penalty": 1.05}, open(os.path.join(path, "generation_config.json"), "w"))
    w = {k: v.to(torch.bfloat16) for k, v in synth_weights(cfg, seed=7).items()}
    w.update(vision_weights_from_fixture())
    keys = sorted(w)
    half = len(keys) // 2                                   # two shards, like a real checkpoint
    save_file({old_key(k): w[k].contiguous() for k in keys[:half]}, os.path.join(path, "model-00001-of-00002.safetensors"))
    save_file({old_key(k): w[k].contiguous() for k in keys[half:]}, os.path.join(path, "model-00002-of-00002.safetensors"))
    tok = Tokenizer(models.WordLevel({wd: i for i, wd in enumerate(WORDS)}, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, unk_token="<unk>", pad_token="<pad>", eos_token="<|im_end|>",
                                   additional_special_tokens=["<|image_pad|>", "<|vision_start|>", "<|vision_end|>", "<|endoftext|>"])
    fast.save_pretrained(path)
    return {"cfg": cfg, "vcfg": vcfg, "weights": w}
