"""Vocabulary-free stand-in for the CLIP tokenizer.

The reference obtains `pipe.tokenizer` from a hub checkpoint (`/root/reference/p2p/edit_syn.py:60`);
no CLIP vocabulary exists offline.  The controllers and the aligner only need
the protocol below, so synthetic runs, tests and the golden-vector generator use this
deterministic tokenizer; a local checkpoint dir configured in `sd_mapping.py` swaps in the real
`transformers.CLIPTokenizer` (see `pipeline.py`).

Protocol used by the path (`/root/reference/p2p/model/sd_utils.py:42-55`, `seq_aligner.py:108-109,
139`):
    tok(prompts, padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
    tok.encode(text) -> [BOS, ..., EOS]     tok.decode([id]) -> piece string
    tok.model_max_length
Words longer than `piece_len` characters split into several tokens, so multi-token words —
which drive `get_word_inds` and the ratio rows of the replacement mapper — are exercised.
"""
import zlib
from types import SimpleNamespace
from typing import List, Union

import torch


class WordPieceTokenizer:
    bos_token_id = 49406
    eos_token_id = 49407
    pad_token_id = 49407
    vocab_size = 49408

    def __init__(self, model_max_length: int = 77, piece_len: int = 6):
        self.model_max_length = model_max_length
        self.piece_len = piece_len
        self._pieces = {}

    def _piece_id(self, piece: str) -> int:
        tid = 1 + zlib.crc32(piece.encode("utf-8")) % (self.bos_token_id - 1)
        while self._pieces.setdefault(tid, piece) != piece:  # open addressing on a clash
            tid = 1 + tid % (self.bos_token_id - 1)
        return tid

    def encode(self, text: str) -> List[int]:
        ids = [self.bos_token_id]
        for word in text.lower().split():
            for k in range(0, len(word), self.piece_len):
                ids.append(self._piece_id(word[k:k + self.piece_len]))
        ids.append(self.eos_token_id)
        return ids

    def decode(self, ids) -> str:
        out = []
        for t in ids:
            t = int(t)
            if t == self.bos_token_id:
                out.append("<|startoftext|>")
            elif t == self.eos_token_id:
                out.append("<|endoftext|>")
            else:
                out.append(self._pieces.get(t, ""))
        return " ".join(out)

    def __call__(self, text: Union[str, List[str]], padding="max_length", max_length=None,
                 truncation=True, return_tensors="pt"):
        texts = [text] if isinstance(text, str) else list(text)
        max_length = max_length or self.model_max_length
        rows = []
        for t in texts:
            ids = self.encode(t)
            if len(ids) > max_length:
                ids = ids[:max_length - 1] + [self.eos_token_id]
            ids = ids + [self.pad_token_id] * (max_length - len(ids))
            rows.append(ids)
        return SimpleNamespace(input_ids=torch.tensor(rows, dtype=torch.long))
