"""Tokenizers.  Real models: the HF ``tokenizer.json`` in the local model
directory, through the ``tokenizers`` library (nothing is downloaded).  When the
directory has none (synthetic-weight benchmarking, tiny test models) text is
mapped byte-wise onto the vocabulary so that the text routes still work; such
output is not language, and the server says so at start-up."""
from __future__ import annotations

import os
from typing import List


class ByteTokenizer:
    """UTF-8 bytes -> ids (byte + offset, folded into the vocabulary)."""
    is_fallback = True

    def __init__(self, vocab_size: int):
        self.vocab_size = vocab_size
        self.offset = 3 if vocab_size >= 259 else 0
        self.bos_token_id = None
        self.eos_token_id = None

    def encode(self, text: str, add_special_tokens: bool = True) -> List[int]:
        ids = [(b + self.offset) % self.vocab_size for b in text.encode("utf-8")]
        return ids or [self.offset % self.vocab_size]

    def decode(self, ids: List[int]) -> str:
        data = bytes(((i - self.offset) % self.vocab_size) % 256 for i in ids)
        return data.decode("utf-8", errors="replace")


class HFTokenizer:
    is_fallback = False

    def __init__(self, path: str):
        from tokenizers import Tokenizer
        self.tk = Tokenizer.from_file(path)
        self.vocab_size = self.tk.get_vocab_size()
        self.bos_token = self.eos_token = None
        self.bos_token_id = self.eos_token_id = None
        self.chat_template = None

    def encode(self, text: str, add_special_tokens: bool = True) -> List[int]:
        """add_special_tokens=False: the tokenizer.json post-processor (the BOS a Llama-3 tokenizer
        prepends) is skipped -- for text that already spells its special tokens out, such as a
        rendered chat template.  Special tokens written in the text are recognised either way."""
        return self.tk.encode(text, add_special_tokens=add_special_tokens).ids

    def token_to_id(self, token):
        return None if token is None else self.tk.token_to_id(token)

    def decode(self, ids: List[int]) -> str:
        return self.tk.decode(list(ids), skip_special_tokens=True)


def load_tokenizer(model: str, vocab_size: int):
    path = os.path.join(model, "tokenizer.json")
    if os.path.isfile(path):
        tk = HFTokenizer(path)
        cfg = os.path.join(model, "tokenizer_config.json")
        if os.path.isfile(cfg):                           # the chat template and the special-token strings it may name
            import json
            with open(cfg) as f:
                meta = json.load(f)
            tk.chat_template = meta.get("chat_template") if isinstance(meta.get("chat_template"), str) else None
            for key in ("bos_token", "eos_token"):
                v = meta.get(key)
                v = v.get("content") if isinstance(v, dict) else v
                setattr(tk, key, v if isinstance(v, str) else None)
                setattr(tk, key + "_id", tk.token_to_id(getattr(tk, key)))
        return tk
    return ByteTokenizer(vocab_size)


class IncrementalDetokenizer:
    """Text delta per new token; holds back bytes of an unfinished UTF-8 sequence.  Each push
    decodes a window of the last few tokens, not the whole sequence (that made a generation of n
    tokens cost n^2 / 2 token decodes on the server's event loop): the window starts a few tokens
    before the text already emitted, so that a tokenizer whose pieces depend on their neighbours
    (leading-space markers) decodes the new tokens in context, and the delta is what the window's
    text grew by."""
    CONTEXT = 5

    def __init__(self, tokenizer):
        self.tk = tokenizer
        self.ids: List[int] = []
        self.prefix = 0        # window start
        self.read = 0          # tokens whose text has been emitted

    def push(self, new_ids: List[int], final: bool = False) -> str:
        self.ids.extend(new_ids)
        before = self.tk.decode(self.ids[self.prefix:self.read]) if self.read > self.prefix else ""
        now = self.tk.decode(self.ids[self.prefix:])
        if now.endswith("\ufffd") and not final:            # an unfinished multi-byte character: wait for the rest
            return ""
        delta = now[len(before):]
        self.read = len(self.ids)
        self.prefix = max(self.read - self.CONTEXT, 0)
        return delta
