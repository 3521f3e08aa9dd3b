"""ctypes binding of ``lib/libcoderag_tok.so`` (``csrc_host/bpe_tokenizer.cpp``): the byte-level BPE tokenizer UniXcoder uses,
from LOCAL ``vocab.json`` / ``merges.txt``, many texts in parallel straight into an int32 matrix.

Replaces ``RobertaTokenizer.tokenize`` + ``convert_tokens_to_ids`` of ``UniXcoder.tokenize``
(``src/lattice/providers/unixcoder_provider.py:105-122``) on the embedding path; same ids as the HF tokenizer
(``tests/test_tokenizer_native.py``).  It fails loudly when the library has not been built."""

from __future__ import annotations

import ctypes as C
import json
import os
from pathlib import Path

import numpy as np

LIB_PATH = Path(os.environ.get("CODERAG_TOK_LIB", Path(__file__).resolve().parent / "lib" / "libcoderag_tok.so"))
ROBERTA_SPECIALS = ("<s>", "<pad>", "</s>", "<unk>", "<mask>")

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError(f"{LIB_PATH} is missing -- build it with code-rag_amd/build.sh")
        L = C.CDLL(str(LIB_PATH))
        pp, ip = C.POINTER(C.c_char_p), C.POINTER(C.c_int32)
        L.crt_create.restype = C.c_void_p
        L.crt_create.argtypes = [C.c_int, pp, ip, C.c_int, pp, pp, C.c_int, pp, ip, C.c_char_p]
        L.crt_destroy.argtypes = [C.c_void_p]
        L.crt_token_to_id.argtypes = [C.c_void_p, C.c_char_p]
        L.crt_token_to_id.restype = C.c_int
        L.crt_encode_batch.argtypes = [C.c_void_p, C.c_int64, pp, C.POINTER(C.c_int64), C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.crt_encode_batch.restype = C.c_int64
        L.crt_count_wordish_ascii.argtypes = [C.c_char_p, C.c_int64]
        L.crt_count_wordish_ascii.restype = C.c_int64
        _lib = L
    return _lib


def count_wordish_ascii(text: str) -> int:
    """Matches of ``\\w+|[^\\w\\s]`` in an ASCII text, -1 when the text is not ASCII (``crt_count_wordish_ascii``)."""
    if not text.isascii():
        return -1
    b = text.encode("ascii")
    return int(lib().crt_count_wordish_ascii(b, len(b)))


def _strs(items):
    arr = (C.c_char_p * max(1, len(items)))()
    for i, s in enumerate(items):
        arr[i] = s if isinstance(s, bytes) else s.encode("utf-8")
    return arr


class NativeBpeTokenizer:
    """``directory`` holds ``vocab.json`` and ``merges.txt``.  ``specials``: added tokens matched in raw text (defaults to
    RoBERTa's five).  ``lstrip``: tokens that also swallow the whitespace before them -- transformers 4.x registers
    ``AddedToken("<mask>", lstrip=True)`` for the slow RobertaTokenizer, 5.x (the oracle available in this image) does not;
    pass ``lstrip=("<mask>",)`` for the former behaviour."""

    def __init__(self, directory: str, specials=ROBERTA_SPECIALS, lstrip=(), threads: int = 0):
        with open(os.path.join(directory, "vocab.json"), encoding="utf-8") as f:
            vocab = json.load(f)
        left, right = [], []
        with open(os.path.join(directory, "merges.txt"), encoding="utf-8") as f:
            for line in f:
                line = line.rstrip("\n")
                if not line or line.startswith("#version"):
                    continue
                a, _, b = line.partition(" ")
                left.append(a)
                right.append(b)
        toks = list(vocab)
        ids = (C.c_int32 * len(toks))(*[int(vocab[t]) for t in toks])
        sp = [s for s in specials if s in vocab]
        ls = (C.c_int32 * max(1, len(sp)))(*[int(s in lstrip) for s in sp])
        self._h = lib().crt_create(len(toks), _strs(toks), ids, len(left), _strs(left), _strs(right), len(sp), _strs(sp), ls, b"<unk>")
        if not self._h:
            raise ValueError("vocab.json lacks some of the 256 byte-level symbols and has no <unk>: not a byte-level BPE vocabulary")
        self.threads = threads
        self.vocab_size = max(vocab.values()) + 1
        self.cls_id, self.pad_id, self.sep_id = vocab["<s>"], vocab["<pad>"], vocab["</s>"]
        eid = vocab.get("<encoder-only>")                   # from the vocabulary, never hard-coded
        if eid is None:
            raise ValueError("vocab.json has no <encoder-only> token: not a UniXcoder vocabulary")
        self.enc_only_id = eid

    def close(self) -> None:
        if getattr(self, "_h", None) and _lib is not None:
            _lib.crt_destroy(self._h)
            self._h = None

    __del__ = close

    def encode_bodies(self, texts, max_body: int = 508):
        """-> (ids int32 [n, max_body] -- row i valid up to min(lens[i], max_body) -- , lens int32 [n] = untruncated counts)."""
        n = len(texts)
        raw = [t.encode("utf-8") for t in texts]
        ptrs = (C.c_char_p * max(1, n))(*raw)
        lens = (C.c_int64 * max(1, n))(*[len(b) for b in raw])
        ids = np.empty((n, max_body), dtype=np.int32)
        out_len = np.zeros((n,), dtype=np.int32)
        lib().crt_encode_batch(self._h, n, ptrs, lens, max_body, ids.ctypes.data, out_len.ctypes.data, self.threads)
        return ids, out_len

    def encode_body(self, text: str) -> list[int]:
        ids, ln = self.encode_bodies([text], max_body=max(1, 4 * len(text.encode("utf-8")) + 8))
        return ids[0, : int(ln[0])].tolist()
