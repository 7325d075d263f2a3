"""Deterministic stand-in tokenizer.

The real VisRAG-Ret tokenizer (`LlamaTokenizerWrapper`, modeling_minicpmv.py:404-438)
needs `tokenizer.model`, which is not on the box.  The model code reads only these
attributes (modeling_minicpmv.py:173-200,247-252,595-609; modeling_visrag_ret.py:73-79):
  im_start, im_end, unk_token, slice_start, slice_end, add_bos_token, bos_id,
  im_start_id, im_end_id, encode(str) -> List[int]
so any object exposing them (this one, or the real wrapper) can be handed to the encoder.
The same ids feed the oracle and the HIP path, so tokenizer fidelity does not affect parity.
"""
from __future__ import annotations

import re
import zlib
from typing import List

_SPECIAL = ["<unk>", "<s>", "</s>", "<image>", "</image>", "<slice>", "</slice>", "\n"]
_SPLIT = re.compile(r"(<unk>|<image>|</image>|<slice>|</slice>|\n|\s+)")


class StandInTokenizer:
    im_start = "<image>"
    im_end = "</image>"
    slice_start = "<slice>"
    slice_end = "</slice>"
    unk_token = "<unk>"
    add_bos_token = True

    def __init__(self, vocab_size: int = 122753):
        assert vocab_size > 64
        self.vocab_size = vocab_size
        self._special = {t: i for i, t in enumerate(_SPECIAL)}

    # ids mirror a llama sentencepiece layout: 0 <unk>, 1 <s>, 2 </s>
    @property
    def unk_id(self) -> int: return 0
    @property
    def bos_id(self) -> int: return 1
    @property
    def eos_id(self) -> int: return 2
    @property
    def im_start_id(self) -> int: return self._special[self.im_start]
    @property
    def im_end_id(self) -> int: return self._special[self.im_end]

    def _word_id(self, w: str) -> int:
        return 16 + zlib.crc32(w.encode("utf-8")) % (self.vocab_size - 16)

    def encode(self, text: str) -> List[int]:
        ids = [self.bos_id] if self.add_bos_token else []
        for piece in _SPLIT.split(text):
            if not piece:
                continue
            if piece in self._special:
                ids.append(self._special[piece])
            elif piece.isspace():
                continue
            else:
                ids.append(self._word_id(piece))
        return ids


class SentencePieceTokenizer:
    """The reference's `LlamaTokenizerWrapper` (modeling_minicpmv.py:404-438: HF's slow LlamaTokenizer over the
    checkpoint's `tokenizer.model` + the marker strings and id properties the model code reads) restated over the
    `sentencepiece` package alone — transformers 5 no longer ships the slow tokenizer the wrapper subclasses (its
    `LlamaTokenizer` has no `sp_model`), so the checkpoint's own wrapper cannot be imported on a current install.

    `encode(text)` follows the slow tokenizer of the reference's pin (transformers 4.40.2, `legacy=True`,
    PreTrainedTokenizer.tokenize + LlamaTokenizer._tokenize + build_inputs_with_special_tokens):
      * the text is cut at the special tokens `<unk>`, `<s>`, `</s>` and at the MiniCPM-V markers (`<image>`, `</image>`,
        `<slice>`, `</slice>`, and `<ref>` / `<box>` / `<quad>` / `<point>` pairs when the model has them): each maps to
        its piece id directly.  The image placeholder is `<image>` + 64 x `<unk>` + `</image>` (modeling_minicpmv.py:
        595-609) and the reference scatters exactly query_num rows between the two markers (:159-166), so no prefix piece
        may appear between the last `<unk>` and `</image>`;
      * every piece of text between them is encoded by sentencepiece on its own, dummy prefix included (the legacy
        behaviour of the slow tokenizer);
      * `bos` is prepended when `add_bos_token` (the checkpoint's default), no `eos`.
    """
    im_start = "<image>"
    im_end = "</image>"
    ref_start, ref_end = "<ref>", "</ref>"
    box_start, box_end = "<box>", "</box>"
    quad_start, quad_end = "<quad>", "</quad>"
    point_start, point_end = "<point>", "</point>"
    slice_start = "<slice>"
    slice_end = "</slice>"

    def __init__(self, model_file: str, add_bos_token: bool = True, add_eos_token: bool = False):
        import sentencepiece as spm
        self.sp_model = spm.SentencePieceProcessor(model_file=model_file)
        self.add_bos_token, self.add_eos_token = bool(add_bos_token), bool(add_eos_token)
        sp = self.sp_model
        self.unk_token = sp.id_to_piece(sp.unk_id())
        self.bos_token = sp.id_to_piece(sp.bos_id()) if sp.bos_id() >= 0 else None
        self.eos_token = sp.id_to_piece(sp.eos_id()) if sp.eos_id() >= 0 else None
        self._special = {t: sp.piece_to_id(t) for t in (self.unk_token, self.bos_token, self.eos_token) if t}
        for t in (self.im_start, self.im_end, self.slice_start, self.slice_end):
            if sp.piece_to_id(t) == sp.unk_id():
                raise ValueError(f"{model_file}: {t!r} is not a piece of this tokenizer model (the MiniCPM-V markers must be)")
        for t in (self.im_start, self.im_end, self.slice_start, self.slice_end, self.ref_start, self.ref_end, self.box_start,
                  self.box_end, self.quad_start, self.quad_end, self.point_start, self.point_end):
            if sp.piece_to_id(t) != sp.unk_id():
                self._special[t] = sp.piece_to_id(t)
        self._split = re.compile("(" + "|".join(re.escape(t) for t in sorted(self._special, key=len, reverse=True)) + ")")

    @classmethod
    def from_pretrained(cls, path: str, **kw) -> "SentencePieceTokenizer":
        """`path`: a checkpoint directory holding `tokenizer.model` (+ optionally tokenizer_config.json with
        add_bos_token / add_eos_token), or the model file itself."""
        import json
        import os
        if os.path.isdir(path):
            cfg = os.path.join(path, "tokenizer_config.json")
            if os.path.exists(cfg):
                with open(cfg) as f:
                    j = json.load(f)
                kw.setdefault("add_bos_token", j.get("add_bos_token", True))
                kw.setdefault("add_eos_token", j.get("add_eos_token", False))
            path = os.path.join(path, "tokenizer.model")
        return cls(path, **kw)

    @property
    def vocab_size(self) -> int: return self.sp_model.get_piece_size()
    @property
    def eos_id(self) -> int: return self.sp_model.eos_id()
    @property
    def bos_id(self) -> int: return self.sp_model.bos_id()
    @property
    def unk_id(self) -> int: return self.sp_model.unk_id()
    @property
    def im_start_id(self) -> int: return self.sp_model.piece_to_id(self.im_start)
    @property
    def im_end_id(self) -> int: return self.sp_model.piece_to_id(self.im_end)

    def encode(self, text: str) -> List[int]:
        ids: List[int] = [self.bos_id] if self.add_bos_token and self.bos_id >= 0 else []
        for piece in self._split.split(text):
            if not piece:
                continue
            if piece in self._special:
                ids.append(self._special[piece])
            else:
                ids.extend(self.sp_model.encode(piece))
        if self.add_eos_token and self.eos_id >= 0:
            ids.append(self.eos_id)
        return ids

    def decode(self, ids) -> str:
        return self.sp_model.decode([int(i) for i in ids])
