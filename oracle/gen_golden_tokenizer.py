"""TEST INFRASTRUCTURE ONLY — pin `visrag_amd.tokenizer.SentencePieceTokenizer` to the reference's tokenizer class.

The checkpoint's tokenizer is `LlamaTokenizerWrapper` (modeling_minicpmv.py:404-438), a subclass of transformers' SLOW
`LlamaTokenizer` (reference pin transformers==4.40.2).  transformers 5 (this image) ships `LlamaTokenizer` on the `tokenizers`
backend only — no `sp_model` — but it still ships the slow machinery the 4.x class was made of: `PreTrainedTokenizer`
(tokenization_python.py: the added-token trie, `tokenize`, `encode`) and `SentencePieceBackend`
(tokenization_utils_sentencepiece.py: `_tokenize` with the `legacy` switch).  This script rebuilds the 4.x slow
`LlamaTokenizer` from those two (bos / eos handling as LlamaTokenizer.build_inputs_with_special_tokens, legacy=True as
MiniCPM-V-2.0's tokenizer_config has it), binds it as `transformers.LlamaTokenizer` BEFORE the reference module is
imported, so that the reference's OWN `LlamaTokenizerWrapper` class statement subclasses it, loads the committed small
sentencepiece model (llama-style: bpe, byte fallback, dummy prefix, identity normaliser; the MiniCPM-V markers declared as
special added tokens, as the checkpoint's tokenizer_config.json does) and records `tokenizer.encode(prompt)` for prompts built
by the reference's own placeholder code (`get_grid_placeholder`, modeling_minicpmv.py:595-609; the single-image
placeholder of `get_slice_image_placeholder`, :241-246).

    python oracle/gen_golden_tokenizer.py       # -> tests/golden/tokenizer/{tokenizer.model, tokenizer_config.json, expected.json}
"""
from __future__ import annotations

import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "tokenizer")
MARKERS = ["<image>", "</image>", "<ref>", "</ref>", "<box>", "</box>", "<quad>", "</quad>", "<point>", "</point>", "<slice>", "</slice>"]


def train_model():
    import sentencepiece as spm
    words = ("revenue table chart figure growth annual report market share total net income page section summary results method "
             "analysis energy policy climate health data model system network design process quality budget forecast region quarter "
             "Represent this query for retrieving relevant documents : what is the of in 2023 ?").split()
    random.seed(0)
    sents = [" ".join(random.choice(words) for _ in range(random.randint(3, 14))) for _ in range(2000)]
    os.makedirs(OUT, exist_ok=True)
    spm.SentencePieceTrainer.train(sentence_iterator=iter(sents), model_prefix=os.path.join(OUT, "tokenizer"), vocab_size=512, model_type="bpe",
                                   unk_id=0, bos_id=1, eos_id=2, pad_id=-1, byte_fallback=True, character_coverage=1.0,
                                   normalization_rule_name="identity", remove_extra_whitespaces=False, add_dummy_prefix=True,
                                   user_defined_symbols=MARKERS, minloglevel=2)
    os.remove(os.path.join(OUT, "tokenizer.vocab"))
    cfg = {"add_bos_token": True, "add_eos_token": False, "legacy": True, "bos_token": "<s>", "eos_token": "</s>", "unk_token": "<unk>",
           "tokenizer_class": "LlamaTokenizerWrapper", "additional_special_tokens": MARKERS}
    with open(os.path.join(OUT, "tokenizer_config.json"), "w") as f:
        json.dump(cfg, f, indent=1)


def slow_llama_class():
    """transformers 4.x's slow LlamaTokenizer, reassembled from the parts transformers 5 still ships."""
    from transformers.tokenization_utils_sentencepiece import SentencePieceBackend

    class LlamaTokenizer(SentencePieceBackend):
        vocab_files_names = {"vocab_file": "tokenizer.model"}
        model_input_names = ["input_ids", "attention_mask"]

        def __init__(self, vocab_file, unk_token="<unk>", bos_token="<s>", eos_token="</s>", pad_token=None, add_bos_token=True,
                     add_eos_token=False, legacy=True, **kwargs):
            self.add_bos_token, self.add_eos_token = add_bos_token, add_eos_token
            kwargs.pop("special_tokens_pattern", None)
            super().__init__(vocab_file=vocab_file, unk_token=unk_token, bos_token=bos_token, eos_token=eos_token, pad_token=pad_token,
                             add_bos_token=add_bos_token, add_eos_token=add_eos_token, legacy=legacy, special_tokens_pattern="none", **kwargs)

        def build_inputs_with_special_tokens(self, token_ids_0, token_ids_1=None):      # tokenization_llama.py (4.40.2): bos + ids + eos
            bos = [self.bos_token_id] if self.add_bos_token else []
            eos = [self.eos_token_id] if self.add_eos_token else []
            out = bos + token_ids_0 + eos
            if token_ids_1 is not None:
                out = out + bos + token_ids_1 + eos
            return out

    return LlamaTokenizer


def main():
    if not os.path.exists(os.path.join(OUT, "tokenizer.model")):
        train_model()
    import transformers
    slow = slow_llama_class()
    import transformers.models.llama as _ml
    import transformers.models.llama.tokenization_llama as _tl
    # before the reference's `from transformers import LlamaTokenizer` (the lazy top-level module resolves the name through these)
    transformers.LlamaTokenizer = _ml.LlamaTokenizer = _tl.LlamaTokenizer = slow
    from oracle import ref_harness
    ref_harness.install_shims()
    from openmatch.modeling.modeling_minicpmv import modeling_minicpmv as R
    assert R.LlamaTokenizerWrapper.__mro__[1] is slow, R.LlamaTokenizerWrapper.__mro__
    cfg = json.load(open(os.path.join(OUT, "tokenizer_config.json")))
    tok = R.LlamaTokenizerWrapper(vocab_file=os.path.join(OUT, "tokenizer.model"), add_bos_token=cfg["add_bos_token"],
                                  add_eos_token=cfg["add_eos_token"], legacy=cfg["legacy"], additional_special_tokens=cfg["additional_special_tokens"])
    assert hasattr(tok, "sp_model") and tok.im_start == "<image>"

    # placeholders exactly as the reference builds them
    image_ph = tok.im_start + tok.unk_token * 64 + tok.im_end
    grid_2x3 = R.get_grid_placeholder(tok, [2, 3], 64)                        # two columns, three rows of slices: "<slice>...</slice>"
    grid_3x3 = R.get_grid_placeholder(tok, [3, 3], 64)
    prefix = "Represent this query for retrieving relevant documents: "
    prompts = {
        "page_single": image_ph + "\n",                                        # one 448 x 448 page, empty text (modeling_visrag_ret.py:73-79)
        "page_sliced_2x3": image_ph + grid_2x3 + "\n",
        "page_sliced_3x3": image_ph + grid_3x3 + "\n",
        "page_with_caption": image_ph + "\n" + "annual report: revenue growth by region",
        "query_plain": prefix + "what is the total net income in 2023?",
        "query_unicode": prefix + "Umsatz 2023 — café 中文",        # byte fallback
        "query_specials_in_text": prefix + "compare <s> markers </s> and <unk> inside text",
        "text_ws": "  two  leading spaces\nand a\ttab ",
        "empty": "",
    }
    exp = {}
    for name, text in prompts.items():
        ids = tok.encode(text)
        exp[name] = {"text": text, "ids": [int(i) for i in ids]}
    ids = exp["page_sliced_2x3"]["ids"]
    assert ids.count(tok.im_start_id) == 7 and ids.count(tok.im_end_id) == 7
    meta = {"bos_id": tok.bos_id, "eos_id": tok.eos_id, "unk_id": tok.unk_id, "im_start_id": tok.im_start_id, "im_end_id": tok.im_end_id,
            "slice_start_id": tok.convert_tokens_to_ids("<slice>"), "slice_end_id": tok.convert_tokens_to_ids("</slice>"),
            "transformers": transformers.__version__, "made_by": "oracle/gen_golden_tokenizer.py: the reference's LlamaTokenizerWrapper over "
            "transformers' PreTrainedTokenizer + SentencePieceBackend (legacy=True)"}
    with open(os.path.join(OUT, "expected.json"), "w") as f:
        json.dump({"meta": meta, "prompts": exp}, f, indent=1, ensure_ascii=True)
    for k, v in exp.items():
        print(k, len(v["ids"]), v["ids"][:12])


if __name__ == "__main__":
    main()
