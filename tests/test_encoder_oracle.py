"""CPU tier for the encoder: the torch-fp32 oracle against fixtures produced from HF RobertaModel
(tests/golden/gen_encoder_goldens.py), and the host-side pieces of the HIP driver that need no GPU."""
import os

import numpy as np
import pytest

import coderag_amd  # noqa: F401
from coderag_amd import encoder as drv
from oracle import encoder as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load_case(name):
    z = np.load(os.path.join(GOLD, f"encoder_{name}.npz"))
    c = [int(v) for v in z["cfg"]]
    cfg = orc.EncoderConfig(vocab_size=c[0], hidden_size=c[1], num_layers=c[2], num_heads=c[3], intermediate_size=c[4],
                            max_position_embeddings=c[5], type_vocab_size=c[6], pad_token_id=c[7], layer_norm_eps=float(z["eps"]))
    return cfg, int(z["seed"]), z["ids"], z["sent"]


# (hfinit / hfln: the same geometry on HF-init statistics and on HF-init matrices with the sharp fixture's biases and LayerNorm
# parameters -- the fixtures name their weight statistic)
@pytest.mark.parametrize("name", ["tiny", "base", "hfinit", "hfln"])
def test_oracle_matches_hf_fixture(name):
    cfg, seed, ids, sent = load_case(name)
    z = np.load(os.path.join(GOLD, f"encoder_{name}.npz"))
    init = str(z["init"]) if "init" in z.files else "sharp"
    got = orc.forward(orc.random_weights(cfg, seed, init=init), cfg, ids)
    assert np.abs(got - sent).max() < 3e-5          # HF RobertaModel fp32 (eager attention) on the same weights


def test_padding_is_invisible_to_real_tokens():
    """Quirk Q1: the padded-batch row equals the single-text result the reference would return."""
    cfg, seed, ids, _ = load_case("tiny")
    w = orc.random_weights(cfg, seed)
    batch = orc.forward(w, cfg, ids)
    n0 = int((ids[0] != cfg.pad_token_id).sum())
    solo = orc.forward(w, cfg, ids[:1, :n0])
    assert np.abs(solo[0] - batch[0]).max() < 2e-5
    wider = np.concatenate([ids, np.full((ids.shape[0], 64), cfg.pad_token_id, ids.dtype)], axis=1)
    assert np.abs(orc.forward(w, cfg, wider) - batch).max() < 2e-5


def test_product_weight_generator_equals_oracle_generator():
    cfg_o = orc.EncoderConfig(vocab_size=300, num_layers=2)
    cfg_p = drv.EncoderConfig(vocab_size=300, num_layers=2)
    a, b = orc.random_weights(cfg_o, 5), drv.synthetic_weights(cfg_p, 5)
    assert a.keys() == b.keys() and all(np.array_equal(a[k], b[k]) for k in a)


def test_tokenize_wrap_and_truncation():
    tok = drv.HashTokenizer(51416)
    ids = drv.wrap_encoder_only(tok, "def hello_world(x):\n    return x + 1", 512)
    assert ids[:3] == [0, tok.enc_only_id, 2] and ids[-1] == 2 and all(i >= 16 for i in ids[3:-1])
    long = drv.wrap_encoder_only(tok, "a b " * 2000, 512)
    assert len(long) == 512 and long[-1] == 2                     # body cut to max_length - 4 (unixcoder_provider.py:112)
    assert drv.wrap_encoder_only(tok, "", 512) == [0, tok.enc_only_id, 2, 2]
    assert drv.wrap_encoder_only(tok, "same text", 64) == drv.wrap_encoder_only(tok, "same text", 64)


def test_flops_formula_matches_survey():
    for L in (8, 128, 512):
        assert drv.flops_per_chunk(L) == 169_869_312 * L + 36_864 * L * L


def test_length_bucketing_plan():
    m = drv.HipUniXcoder.__new__(drv.HipUniXcoder)
    lengths = [500, 10, 64, 65, 300, 12, 512, 70]
    batches = m.plan_batches(lengths, max_tokens=1024)
    seen = sorted(i for rows, _ in batches for i in rows)
    assert seen == list(range(len(lengths)))
    for rows, L in batches:
        assert L % 16 == 0 and L >= max(lengths[i] for i in rows) and L * len(rows) <= 1024
    assert m.plan_batches([], 1024) == []


@pytest.mark.parametrize("init", ["sharp", "hf"])
def test_the_folded_layernorm_form_is_the_same_function(init):
    """Round 5: the HIP path folds each LayerNorm into the GEMMs around it (rstd (r @ (w gain)^T - mu colsum) + (b + w @ beta) for the
    consumer; the previous LayerNorm's output worked out on the fly for the producer's residual).  ``ln_fold=True`` restates the
    forward that way: in f32 it must be the plain forward up to rounding (modeling_roberta.py:329-340, 387-398 are unchanged
    arithmetic), and at the kernels' storage precision it must sit as close to the fp32 forward as the plain bf16 model does."""
    import numpy as np
    from oracle import encoder as orc
    cfg = orc.EncoderConfig(vocab_size=600, hidden_size=768, num_layers=3, num_heads=12, intermediate_size=3072, max_position_embeddings=130)
    w = orc.random_weights(cfg, 5, init=init)
    ids = orc.synthetic_ids(cfg, [20, 64, 37, 9], 3, pad_to=64)
    plain = orc.forward(w, cfg, ids)
    folded = orc.forward(w, cfg, ids, ln_fold=True)
    assert np.abs(plain - folded).max() <= 1e-5 * np.abs(plain).max()

    def rel(x):
        return (np.linalg.norm(x - plain, axis=1) / np.linalg.norm(plain, axis=1)).max()
    a, b = rel(orc.forward(w, cfg, ids, bf16_storage=True)), rel(orc.forward(w, cfg, ids, bf16_storage=True, ln_fold=True))
    assert b <= 1.25 * a, (a, b)
