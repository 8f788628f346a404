"""Self-contained CLIP byte-pair tokenizer.

The reference leans on transformers 4.27's slow `CLIPTokenizer._tokenize` (utils/richtext_utils.py:150,160,171,195,220),
which later transformers releases dropped.  This is the published CLIP BPE algorithm (OpenAI `simple_tokenizer.py`; the
HF slow tokenizer without ftfy lower-cases and collapses whitespace) over a checkpoint's own `vocab.json` + `merges.txt`
- neither file is available offline, so tests use a synthetic vocabulary and, where transformers can build a tokenizer
from the same files, compare against it.  [memory] parity unpinned against transformers 4.27.
"""
import json
import os
from functools import lru_cache
from types import SimpleNamespace

import regex
import torch

_PAT = regex.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+", regex.IGNORECASE)


@lru_cache()
def _byte_alphabet():
    """GPT-2 reversible byte<->printable-unicode table: printable latin-1 bytes map to themselves, the rest to 256+k."""
    keep = list(range(ord('!'), ord('~') + 1)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    table, extra = {}, 0
    for b in range(256):
        if b in keep:
            table[b] = chr(b)
        else:
            table[b] = chr(256 + extra)
            extra += 1
    return table


class ClipBPETokenizer:
    model_max_length = 77

    def __init__(self, vocab_file, merges_file, pad_token=None, bos_token="<|startoftext|>", eos_token="<|endoftext|>"):
        self.encoder = json.load(open(vocab_file, encoding="utf-8"))
        self.decoder = {v: k for k, v in self.encoder.items()}
        lines = open(merges_file, encoding="utf-8").read().strip().split("\n")[1:49152 - 256 - 2 + 1]
        self.ranks = {tuple(l.split()): i for i, l in enumerate(lines) if l}
        self.bos_token, self.eos_token = bos_token, eos_token
        self.pad_token = pad_token if pad_token is not None else eos_token
        self.bos_token_id = self.encoder[bos_token]
        self.eos_token_id = self.encoder[eos_token]
        self.pad_token_id = self.encoder[self.pad_token]
        self._cache = {bos_token: bos_token, eos_token: eos_token}

    @classmethod
    def from_pretrained(cls, path, subfolder=None):
        d = os.path.join(path, subfolder) if subfolder else path
        pad = None
        sp = os.path.join(d, "special_tokens_map.json")
        if os.path.exists(sp):
            pt = json.load(open(sp)).get("pad_token")
            pad = pt.get("content") if isinstance(pt, dict) else pt
        return cls(os.path.join(d, "vocab.json"), os.path.join(d, "merges.txt"), pad_token=pad)

    def _bpe(self, token):
        if token in self._cache:
            return self._cache[token]
        word = list(token[:-1]) + [token[-1] + "</w>"]
        while len(word) > 1:
            best = min(zip(word, word[1:]), key=lambda p: self.ranks.get(p, float("inf")))
            if best not in self.ranks:
                break
            a, b = best
            merged, i = [], 0
            while i < len(word):
                if i + 1 < len(word) and word[i] == a and word[i + 1] == b:
                    merged.append(a + b)
                    i += 2
                else:
                    merged.append(word[i])
                    i += 1
            word = merged
        out = " ".join(word)
        self._cache[token] = out
        return out

    def _tokenize(self, text):
        text = " ".join(text.split()).strip().lower()
        alphabet = _byte_alphabet()
        pieces = []
        for tok in regex.findall(_PAT, text):
            tok = "".join(alphabet[b] for b in tok.encode("utf-8"))
            pieces.extend(self._bpe(tok).split(" "))
        return pieces

    tokenize = _tokenize

    def convert_tokens_to_ids(self, tokens):
        unk = self.encoder[self.eos_token]
        return [self.encoder.get(t, unk) for t in tokens]

    def encode(self, text, max_length=None, truncation=False, padding=None):
        ids = [self.bos_token_id] + self.convert_tokens_to_ids(self._tokenize(text)) + [self.eos_token_id]
        max_length = max_length or self.model_max_length
        if truncation and len(ids) > max_length:
            ids = ids[:max_length - 1] + [self.eos_token_id]
        if padding == "max_length":
            ids = ids + [self.pad_token_id] * (max_length - len(ids))
        return ids

    def __call__(self, text, padding=None, max_length=None, truncation=False, return_tensors=None, **_):
        texts = [text] if isinstance(text, str) else list(text)
        rows = [self.encode(t, max_length, truncation, padding) for t in texts]
        if return_tensors == "pt":
            width = max(len(r) for r in rows)
            rows = [r + [self.pad_token_id] * (width - len(r)) for r in rows] if padding else rows
            return SimpleNamespace(input_ids=torch.tensor(rows, dtype=torch.long))
        return SimpleNamespace(input_ids=rows if not isinstance(text, str) else rows[0])
