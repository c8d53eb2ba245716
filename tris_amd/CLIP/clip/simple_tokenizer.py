"""CLIP byte-level BPE tokenizer (host side of CLIP/clip/simple_tokenizer.py:62-133 in the reference).

The merge table `bpe_simple_vocab_16e6.txt.gz` is OpenAI CLIP's vocabulary DATA file (MIT licence, the same file every CLIP
distribution carries); it ships next to this module so that `clip.tokenize` / ReferDataset work on a machine without a
reference checkout.  Lookup order: $TRIS_BPE_VOCAB, next to this module, a reference checkout's CLIP/clip/.
"""
import gzip
import html
import os
from functools import lru_cache

import regex as re

try:  # the reference uses ftfy.fix_text; identity for plain ASCII/UTF-8 text
    import ftfy
    _fix = ftfy.fix_text
except Exception:  # pragma: no cover
    def _fix(s):
        return s

VOCAB_NAME = "bpe_simple_vocab_16e6.txt.gz"


def default_bpe():
    cands = [os.environ.get("TRIS_BPE_VOCAB"), os.path.join(os.path.dirname(os.path.abspath(__file__)), VOCAB_NAME),
             os.path.join("/root/reference/CLIP/clip", VOCAB_NAME)]
    for c in cands:
        if c and os.path.exists(c):
            return c
    raise FileNotFoundError(f"{VOCAB_NAME} not found; set TRIS_BPE_VOCAB to its path")


@lru_cache()
def byte_alphabet():
    """Reversible byte -> printable unicode char map (GPT-2 convention)."""
    keep = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
    table, extra = {}, 0
    for b in range(256):
        if b in keep:
            table[b] = chr(b)
        else:
            table[b] = chr(256 + extra)
            extra += 1
    return table


class SimpleTokenizer:
    def __init__(self, bpe_path=None):
        bpe_path = bpe_path or default_bpe()
        alphabet = byte_alphabet()
        lines = gzip.open(bpe_path).read().decode("utf-8").split("\n")
        merges = [tuple(l.split()) for l in lines[1:49152 - 256 - 2 + 1]]
        # vocabulary order: bytes in the keep-then-extra order, the same with </w>, merges, specials
        order = [alphabet[b] for b in (list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256)))]
        order += [alphabet[b] for b in range(256) if alphabet[b] not in order]
        vocab = order + [c + "</w>" for c in order] + ["".join(m) for m in merges]
        vocab += ["<|startoftext|>", "<|endoftext|>"]
        self.byte_encoder = alphabet
        self.byte_decoder = {v: k for k, v in alphabet.items()}
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.decoder = {i: tok for tok, i in self.encoder.items()}
        self.rank = {m: i for i, m in enumerate(merges)}
        self.cache = {"<|startoftext|>": "<|startoftext|>", "<|endoftext|>": "<|endoftext|>"}
        self.pat = re.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
                              re.IGNORECASE)

    def bpe(self, token):
        if token in self.cache:
            return self.cache[token]
        word = list(token[:-1]) + [token[-1] + "</w>"]
        while len(word) > 1:
            pairs = [(self.rank.get((a, b), float("inf")), i) for i, (a, b) in enumerate(zip(word, word[1:]))]
            best, _ = min(pairs)
            if best == float("inf"):
                break
            a, b = next((x, y) for x, y in zip(word, word[1:]) if self.rank.get((x, y)) == best)
            merged, i = [], 0
            while i < len(word):
                if i < len(word) - 1 and word[i] == a and word[i + 1] == b:
                    merged.append(a + b)
                    i += 2
                else:
                    merged.append(word[i])
                    i += 1
            word = merged
        out = " ".join(word)
        self.cache[token] = out
        return out

    def encode(self, text):
        text = html.unescape(html.unescape(_fix(text))).strip()
        text = re.sub(r"\s+", " ", text).strip().lower()
        ids = []
        for tok in re.findall(self.pat, text):
            tok = "".join(self.byte_encoder[b] for b in tok.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self.bpe(tok).split(" "))
        return ids

    def decode(self, tokens):
        text = "".join(self.decoder[t] for t in tokens)
        return bytearray(self.byte_decoder[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")
