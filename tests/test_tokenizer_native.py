"""The native byte-level BPE tokenizer (csrc_host/bpe_tokenizer.cpp) against the HF implementation the reference uses
(transformers RobertaTokenizer over the `tokenizers` library): identical ids on source files, on whitespace / unicode /
contraction / special-token corner cases and on random strings, for a vocabulary trained here (no hub access)."""
import glob
import os
import random

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SPECIALS = ["<s>", "<pad>", "</s>", "<unk>", "<mask>", "<encoder-only>", "<decoder-only>", "<encoder-decoder>"]


@pytest.fixture(scope="module")
def toks(tmp_path_factory):
    from tokenizers import ByteLevelBPETokenizer
    from transformers import RobertaTokenizer
    import coderag_amd  # noqa: F401
    from coderag_amd.tokenizer_native import NativeBpeTokenizer
    d = str(tmp_path_factory.mktemp("vocab"))
    files = sorted(glob.glob(os.path.join(ROOT, "code-rag_amd", "**", "*.py"), recursive=True) + glob.glob(os.path.join(ROOT, "tests", "*.py")))
    tr = ByteLevelBPETokenizer(add_prefix_space=False)
    tr.train(files, vocab_size=6000, min_frequency=2, special_tokens=SPECIALS)
    tr.save_model(d)
    return RobertaTokenizer(os.path.join(d, "vocab.json"), os.path.join(d, "merges.txt")), NativeBpeTokenizer(d), files


def _hf_ids(hf, text):
    return hf.convert_tokens_to_ids(hf.tokenize(text))


def test_same_ids_on_source_files(toks):
    hf, nat, files = toks
    chunks = []
    for f in files:
        s = open(f, encoding="utf-8").read()
        chunks += [s[i:i + 1500] for i in range(0, len(s), 1500)]
    ids, lens = nat.encode_bodies(chunks, max_body=4096)
    assert int(lens.max()) <= 4096
    bad = [i for i, c in enumerate(chunks) if ids[i, : lens[i]].tolist() != _hf_ids(hf, c)]
    assert not bad, f"{len(bad)} of {len(chunks)} chunks differ, first: {chunks[bad[0]][:80]!r}"


CASES = ["", " ", "  ", "   x", "x   ", "a\n\n\nb", "\t\tif x:\r\n\t\t\treturn  y\n", "it's they're I'll we'd I've I'm don't 'sx 'S",
         "naïve café ÀÉÎõü ß ẞ ǅ Ω", "日本語のコメント と 中文注释", "한국어 텍스트 123", "emoji 😀👍🏽 flags 🇩🇪", "x²+y³=z¼ ①②③ ٣٤٥ १२३",
         "a\u00a0b\u2003c\u3000d\u2028e\u0085f", "zero\u200bwidth\u200djoiner", "tab\there  \t mixed \n \n", "!!!???...,,,;;; ((([[[{{{",
         "snake_case camelCase PascalCase kebab-case SCREAMING_CASE x1y2z3 0x1F 1e-9 3.14", "<s>in text</s> and <pad><unk> <mask> here  <mask>x",
         "a<mask>", "<mask>", "   <mask>", "<encoder-only> marker <decoder-only>", "< s > <s > <mask", "\\n \\t \"quoted\" 'single' `tick`",
         "\x00\x01\x7f control", "end with spaces   ", "\n", "\n\n", " \n ", "x\n", "ｆｕｌｌｗｉｄｔｈ ＡＢＣ １２３", "combining e\u0301 a\u030a n\u0303"]


def test_same_ids_on_corner_cases(toks):
    hf, nat, _ = toks
    for c in CASES:
        assert nat.encode_body(c) == _hf_ids(hf, c), repr(c)


def test_same_ids_on_random_strings(toks):
    hf, nat, _ = toks
    rng = random.Random(3)
    alphabet = list("abcXYZ019 _-+*/=<>()[]{}'\"\\\n\t.,:;!?#@$%^&|~`") + ["é", "ß", "日", "本", "😀", "\u00a0", "\u2003", "١", "²", "ǅ", "'s", "'re", "<s>", "<mask>", "  "]
    texts = ["".join(rng.choice(alphabet) for _ in range(rng.randrange(0, 60))) for _ in range(3000)]
    ids, lens = nat.encode_bodies(texts, max_body=512)
    for i, t in enumerate(texts):
        assert ids[i, : lens[i]].tolist() == _hf_ids(hf, t), repr(t)


def test_truncation_and_threads(toks):
    hf, nat, files = toks
    text = open(files[0], encoding="utf-8").read()
    full = _hf_ids(hf, text)
    ids, lens = nat.encode_bodies([text, "x", text], max_body=37)
    assert lens.tolist() == [len(full), 1, len(full)]
    assert ids[0].tolist() == full[:37] and ids[2].tolist() == full[:37] and ids[1, 0] == _hf_ids(hf, "x")[0]
    one = type(nat).__new__(type(nat))
    one.__dict__.update(nat.__dict__)
    one.threads = 1
    a, la = one.encode_bodies([text] * 40, max_body=64)
    nat.threads = 8
    b, lb = nat.encode_bodies([text] * 40, max_body=64)
    one._h = None                                                   # (shared handle: only `nat` frees it)
    assert np.array_equal(a, b) and np.array_equal(la, lb)


def test_guards_against_foreign_vocabularies(tmp_path):
    """A vocabulary that cannot express every byte (and has no <unk>) is refused at load; token ids outside the checkpoint's
    embedding table are refused before anything reaches the device."""
    import json
    from types import SimpleNamespace as NS
    import coderag_amd  # noqa: F401
    from coderag_amd.encoder import EncoderConfig, HipUniXcoder
    from coderag_amd.tokenizer_native import NativeBpeTokenizer
    json.dump({"<s>": 0, "<pad>": 1, "</s>": 2, "<encoder-only>": 3, "a": 4, "b": 5}, open(tmp_path / "vocab.json", "w"))
    open(tmp_path / "merges.txt", "w").write("#version: 0.2\na b\n")
    with pytest.raises(ValueError, match="byte-level"):
        NativeBpeTokenizer(str(tmp_path))
    fake = NS(cfg=EncoderConfig(vocab_size=100))
    HipUniXcoder._check_ids(fake, np.asarray([[5, 99, 7]], np.int32), np.asarray([3]))
    HipUniXcoder._check_ids(fake, np.asarray([[5, 99, 1000]], np.int32), np.asarray([2]))      # the 1000 is beyond the row's length
    with pytest.raises(ValueError, match="embedding table"):
        HipUniXcoder._check_ids(fake, np.asarray([[5, 100, 7]], np.int32), np.asarray([3]))
    with pytest.raises(ValueError, match="embedding table"):
        HipUniXcoder._check_ids(fake, np.asarray([[-1, 3, 7]], np.int32), np.asarray([3]))


def test_same_ids_on_random_code_points(toks):
    """Strings of code points drawn from the whole Unicode range (every general category, astral planes, unassigned points;
    surrogates excluded -- they are not text), mixed with ASCII letters / digits / spaces so that the pre-tokenizer's
    letter / number / other / whitespace classes meet at every kind of boundary."""
    hf, nat, _ = toks
    rng = random.Random(11)
    ascii_mix = list("ab Z09 _.\n\t'")

    def point():
        while True:
            r = rng.random()
            c = rng.randrange(0x80, 0x3000) if r < 0.5 else rng.randrange(0x3000, 0x10000) if r < 0.8 else rng.randrange(0x10000, 0x110000)
            if not 0xD800 <= c <= 0xDFFF:
                return chr(c)
    texts = ["".join(point() if rng.random() < 0.6 else rng.choice(ascii_mix) for _ in range(rng.randrange(1, 24))) for _ in range(4000)]
    ids, lens = nat.encode_bodies(texts, max_body=512)
    bad = [t for i, t in enumerate(texts) if ids[i, :lens[i]].tolist() != _hf_ids(hf, t)]
    assert not bad, [(repr(t), [hex(ord(ch)) for ch in t]) for t in bad[:3]]
