#!/usr/bin/env python3
"""Pin oracle/encoder.py against the installed HF RobertaModel and freeze small fixtures.

Survey-container-only (needs `transformers`; never imported by the package or the tests).  The HF model is built
locally from a RobertaConfig (no hub access) with is_decoder=False and a 2-D attention mask -- the semantics the
reference's 3-D mask had under transformers 4.x -- and loaded with the seeded weights of
oracle.encoder.random_weights.  Writes tests/golden/encoder_{tiny,base,hfinit,hfln}.npz: config, seed, ids and the expected
sentence embeddings (HF fp32).  Weights are NOT stored: they are regenerated from the seed.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import encoder as enc  # noqa: E402


def hf_forward(cfg: enc.EncoderConfig, weights, ids):
    from transformers import RobertaConfig, RobertaModel
    hc = RobertaConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_layers,
                       num_attention_heads=cfg.num_heads, intermediate_size=cfg.intermediate_size,
                       max_position_embeddings=cfg.max_position_embeddings, type_vocab_size=cfg.type_vocab_size,
                       layer_norm_eps=cfg.layer_norm_eps, pad_token_id=cfg.pad_token_id, hidden_act="gelu",
                       hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, is_decoder=False)
    hc._attn_implementation = "eager"
    model = RobertaModel(hc, add_pooling_layer=False).eval()
    missing, unexpected = model.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()}, strict=False)
    assert not unexpected and all("position_ids" in m or "token_type_ids" in m for m in missing), (missing, unexpected)
    ids_t = torch.from_numpy(ids)
    mask = ids_t.ne(cfg.pad_token_id)
    with torch.no_grad():
        tok = model(ids_t, attention_mask=mask.long())[0]
        sent = (tok * mask.unsqueeze(-1)).sum(1) / mask.sum(-1).unsqueeze(-1)       # unixcoder_provider.py:152-154
    return sent.numpy(), tok.numpy()


def case(name, cfg, seed, lengths, pad_to, init="sharp"):
    w = enc.random_weights(cfg, seed, init=init)
    ids = enc.synthetic_ids(cfg, lengths, seed + 1, pad_to=pad_to)
    if name == "tiny":
        ids[1, 5] = cfg.pad_token_id          # an interior pad token: masked as key and in the pool, like the reference's ids.ne(pad)
    hs, ht = hf_forward(cfg, w, ids)
    os_, ot = enc.forward(w, cfg, ids, return_tokens=True)
    m = ids != cfg.pad_token_id
    err_s = np.abs(hs - os_).max()
    err_t = np.abs((ht - ot)[m]).max()
    print(f"{name}: oracle vs HF  max|d sent|={err_s:.2e}  max|d tok(valid)|={err_t:.2e}  |sent|~{np.abs(hs).mean():.3f}")
    assert err_s < 2e-5 and err_t < 1e-4
    # pad invariance (quirk Q1): the single-text result equals the padded-batch row
    solo = enc.forward(w, cfg, ids[:1, : lengths[0]])
    assert np.abs(solo[0] - os_[0]).max() < 2e-5
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"encoder_{name}.npz"), seed=seed, ids=ids, sent=hs, init=init,
                        cfg=np.array([cfg.vocab_size, cfg.hidden_size, cfg.num_layers, cfg.num_heads, cfg.intermediate_size,
                                      cfg.max_position_embeddings, cfg.type_vocab_size, cfg.pad_token_id]),
                        eps=cfg.layer_norm_eps)


if __name__ == "__main__":
    torch.manual_seed(0)
    tiny = enc.EncoderConfig(vocab_size=1000, num_layers=2)
    case("tiny", tiny, 11, [17, 64, 9, 40], 64)
    case("base", enc.EncoderConfig(), 23, [128, 33, 77], 128)
    # the same geometry with HF-init-like statistics (N(0, 0.02^2) matrices, zero biases, unit LayerNorm): real checkpoints sit
    # between this and the deliberately sharp "base" weights
    case("hfinit", enc.EncoderConfig(), 29, [128, 33, 77, 200], 208, init="hf")
    # between the two (round 5): HF-init matrices with the sharp fixture's biases and LayerNorm gains / biases
    if "hfln" in sys.argv[1:] or not sys.argv[1:]:
        case("hfln", enc.EncoderConfig(), 31, [96, 40, 160, 12], 160, init="hf_ln")
