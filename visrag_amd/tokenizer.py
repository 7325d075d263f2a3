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
