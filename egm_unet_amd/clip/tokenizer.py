"""Host-side byte-pair-encoding tokenizer for the CLIP text encoder: the counterpart of clip.tokenize
(clip/clip.py:313-353, context_length=248 for Long-CLIP) and of the OpenAI BPE it drives.

Own implementation of the published CLIP BPE scheme: UTF-8 bytes are mapped to printable code points, words are split by
the CLIP pattern, and adjacent symbol pairs are merged greedily by merge rank until no ranked pair is left.  The merge
table is the data file data/bpe_simple_vocab_16e6.txt.gz (OpenAI CLIP vocabulary, 48,894 merges -> 49,408 tokens).
`ftfy` is not available here: text is HTML-unescaped and whitespace-normalised only (identical for clean UTF-8 input).
"""
import gzip
import html
import os
from typing import List, Union

import regex
import torch

_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "bpe_simple_vocab_16e6.txt.gz")
_N_MERGES = 49152 - 256 - 2
_SOT, _EOT = "<|startoftext|>", "<|endoftext|>"
_END = "</w>"


def _byte_alphabet():
    """256 bytes -> 256 distinct printable code points (printable Latin-1 maps to itself, the rest to U+0100...)."""
    keep = [b for b in range(256) if 33 <= b <= 126 or 161 <= b <= 172 or 174 <= b <= 255]
    table, extra = {}, 0
    for b in keep:
        table[b] = chr(b)
    for b in range(256):
        if b not in table:
            table[b] = chr(256 + extra)
            extra += 1
    return keep, table


class BPETokenizer:
    def __init__(self, path: str = _DATA):
        keep, self.byte_map = _byte_alphabet()
        order = keep + [b for b in range(256) if b not in set(keep)]          # vocabulary order of the byte symbols
        with gzip.open(path) as f:
            lines = f.read().decode("utf-8").split("\n")
        merges = [tuple(l.split()) for l in lines[1:_N_MERGES + 1]]
        symbols = [self.byte_map[b] for b in order]
        vocab = symbols + [s + _END for s in symbols] + ["".join(m) for m in merges] + [_SOT, _EOT]
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.decoder = {i: tok for tok, i in self.encoder.items()}
        self.rank = {m: i for i, m in enumerate(merges)}
        self.sot, self.eot = self.encoder[_SOT], self.encoder[_EOT]
        self.splitter = regex.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
                                      regex.IGNORECASE)
        self._memo = {_SOT: [_SOT], _EOT: [_EOT]}

    def _merge_word(self, word: str) -> List[str]:
        """Greedy lowest-rank-first merging of one pre-token (already mapped to the byte alphabet)."""
        hit = self._memo.get(word)
        if hit is not None:
            return hit
        parts = list(word[:-1]) + [word[-1] + _END]
        while len(parts) > 1:
            best, best_rank = None, None
            for a, b in zip(parts, parts[1:]):
                r = self.rank.get((a, b))
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = (a, b), r
            if best is None:
                break
            a, b = best
            out, i = [], 0
            while i < len(parts):
                if i + 1 < len(parts) and parts[i] == a and parts[i + 1] == b:
                    out.append(a + b); i += 2
                else:
                    out.append(parts[i]); i += 1
            parts = out
        self._memo[word] = parts
        return parts

    def encode(self, text: str) -> List[int]:
        text = html.unescape(html.unescape(text)).strip()
        text = regex.sub(r"\s+", " ", text).strip().lower()
        ids = []
        for tok in self.splitter.findall(text):
            mapped = "".join(self.byte_map[b] for b in tok.encode("utf-8"))
            ids.extend(self.encoder[p] for p in self._merge_word(mapped))
        return ids

    def decode(self, ids) -> str:
        inv = {c: b for b, c in self.byte_map.items()}
        text = "".join(self.decoder[int(i)] for i in ids)
        return bytearray(inv[c] for c in text.replace(_END, " ") if c in inv).decode("utf-8", errors="replace")


_tokenizer = None


def tokenize(texts: Union[str, List[str]], context_length: int = 77 * 4 - 60, truncate: bool = False) -> torch.Tensor:
    """-> int32 [len(texts), context_length]: [SOT] + BPE ids + [EOT], zero padded; with truncate=True an over-long text is
    cut to context_length and its last token forced to EOT, otherwise it raises (clip/clip.py:313-353)."""
    global _tokenizer
    if _tokenizer is None:
        _tokenizer = BPETokenizer()
    if isinstance(texts, str):
        texts = [texts]
    out = torch.zeros(len(texts), context_length, dtype=torch.int32)
    for i, t in enumerate(texts):
        ids = [_tokenizer.sot] + _tokenizer.encode(t) + [_tokenizer.eot]
        if len(ids) > context_length:
            if not truncate:
                raise RuntimeError(f"Input {t} is too long for context length {context_length}")
            ids = ids[:context_length]
            ids[-1] = _tokenizer.eot
        out[i, :len(ids)] = torch.tensor(ids, dtype=torch.int32)
    return out
