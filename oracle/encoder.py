"""CPU oracle of the UniXcoder encoder forward -- TEST INFRASTRUCTURE ONLY.

Plain-torch fp32 restatement of what ``UniXcoder.forward`` computes
(``src/lattice/providers/unixcoder_provider.py:137-155``) through HF ``RobertaModel``
(installed ``transformers/models/roberta/modeling_roberta.py``: embeddings :56-121, self-attention
:158-183, output/LN :329-340, FFN :372-398): post-LN encoder, bidirectional attention with
key-padding masking (quirk Q2: NOT causal), erf-GELU, masked mean pooling without L2 normalisation.

Pinned in this container against ``transformers.RobertaModel`` itself on seeded random weights
(tests/golden/gen_encoder_goldens.py -> tests/golden/encoder_*.npz).  No UniXcoder checkpoint exists offline,
so parity with the *published weights* is unpinned; parity of the *computation* is pinned.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""

from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch


@dataclass(frozen=True)
class EncoderConfig:
    """RoBERTa-base geometry = microsoft/unixcoder-base (values recalled from its public config.json; a local
    checkpoint's config.json overrides them when one is supplied)."""

    vocab_size: int = 51416
    hidden_size: int = 768
    num_layers: int = 12
    num_heads: int = 12
    intermediate_size: int = 3072
    max_position_embeddings: int = 1026
    type_vocab_size: int = 10
    layer_norm_eps: float = 1e-5
    pad_token_id: int = 1


def random_weights(cfg: EncoderConfig, seed: int, init: str = "sharp") -> dict[str, np.ndarray]:
    """Seeded weights under HF ``RobertaModel`` state-dict names.  numpy's Generator stream is identical on every
    machine, so the GPU box regenerates these bit-for-bit.  Scales are chosen so activations stay O(1) through
    12 post-LN layers (std 0.02 like HF init would make attention nearly uniform and hide softmax bugs).
    ``init="hf"`` gives the statistics of HF's ``_init_weights`` instead (matrices and tables N(0, 0.02^2), biases 0, LayerNorm
    weight 1 / bias 0): the second fixture, on which the bf16 pipeline's deviation is reported beside the sharp one."""
    rng = np.random.default_rng(seed)
    H, F = cfg.hidden_size, cfg.intermediate_size

    hf = init in ("hf", "hf_ln")          # matrices and tables N(0, 0.02^2)
    hf_vec = init == "hf"                 # ... and zero biases / unit LayerNorm; "hf_ln" keeps the sharp fixture's biases and LayerNorm gains
    if init not in ("sharp", "hf", "hf_ln"):
        raise ValueError(f"unknown init {init!r}")

    def mat(n, k, std):
        std = 0.02 if hf else std
        return (rng.standard_normal((n, k), dtype=np.float32) * np.float32(std))

    def vec(n, std, mean=0.0):
        if hf_vec:   # (the draw is still made, so all flavours consume the generator identically)
            return rng.standard_normal(n, dtype=np.float32) * np.float32(0.0) + np.float32(mean)
        return (rng.standard_normal(n, dtype=np.float32) * np.float32(std) + np.float32(mean))
    w = {
        "embeddings.word_embeddings.weight": mat(cfg.vocab_size, H, 0.5),
        "embeddings.position_embeddings.weight": mat(cfg.max_position_embeddings, H, 0.3),
        "embeddings.token_type_embeddings.weight": mat(cfg.type_vocab_size, H, 0.1),
        "embeddings.LayerNorm.weight": vec(H, 0.1, 1.0),
        "embeddings.LayerNorm.bias": vec(H, 0.05),
    }
    w["embeddings.word_embeddings.weight"][cfg.pad_token_id] = 0.0          # nn.Embedding padding_idx rows are zero
    w["embeddings.position_embeddings.weight"][cfg.pad_token_id] = 0.0
    for i in range(cfg.num_layers):
        p = f"encoder.layer.{i}."
        for name in ("query", "key", "value"):
            w[p + f"attention.self.{name}.weight"] = mat(H, H, 2.0 / math.sqrt(H))
            w[p + f"attention.self.{name}.bias"] = vec(H, 0.05)
        w[p + "attention.output.dense.weight"] = mat(H, H, 1.0 / math.sqrt(H))
        w[p + "attention.output.dense.bias"] = vec(H, 0.05)
        w[p + "attention.output.LayerNorm.weight"] = vec(H, 0.1, 1.0)
        w[p + "attention.output.LayerNorm.bias"] = vec(H, 0.05)
        w[p + "intermediate.dense.weight"] = mat(F, H, 1.0 / math.sqrt(H))
        w[p + "intermediate.dense.bias"] = vec(F, 0.05)
        w[p + "output.dense.weight"] = mat(H, F, 1.0 / math.sqrt(F))
        w[p + "output.dense.bias"] = vec(H, 0.05)
        w[p + "output.LayerNorm.weight"] = vec(H, 0.1, 1.0)
        w[p + "output.LayerNorm.bias"] = vec(H, 0.05)
    return w


def position_ids(ids: torch.Tensor, pad: int) -> torch.Tensor:
    """modeling_roberta.py create_position_ids_from_input_ids: cumsum(ids != pad) * (ids != pad) + pad."""
    mask = ids.ne(pad).to(torch.int64)
    return torch.cumsum(mask, dim=1) * mask + pad


def _ln(x: torch.Tensor, g: torch.Tensor, b: torch.Tensor, eps: float) -> torch.Tensor:
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * g + b


def forward(weights: dict[str, np.ndarray], cfg: EncoderConfig, ids: np.ndarray, return_tokens: bool = False,
            dtype: torch.dtype = torch.float32, bf16_storage: bool = False, device: str = "cpu", ln_fold: bool = False,
            residual_f32: bool = False):
    """ids: int [B, L], right- or arbitrarily padded with cfg.pad_token_id.  Returns sentence embeddings [B, H]
    (and the token embeddings [B, L, H]).  unixcoder_provider.py:146-155.

    ``bf16_storage=True`` keeps the arithmetic in f32 but rounds to bf16 exactly where the HIP path STORES bf16
    (matrix weights and embedding tables, every activation written between kernels, and the softmax probabilities
    fed to the P.V product): the model of "the same computation at the kernels' storage precision" that the GPU
    parity test compares against.  Biases and LayerNorm parameters stay f32 there, as in the kernels.

    ``ln_fold=True`` evaluates the post-LN layers the way the HIP path does since round 5 (csrc/crh_encoder.hip, "LayerNorm folded
    into the GEMMs around it"): the residual stream stays un-normalised, a consumer GEMM multiplies it by gain-scaled weights and
    finishes the normalisation per element (``rstd * acc + nmr * colsum + folded bias``), a producer GEMM adds the previous
    LayerNorm's output worked out on the fly.  Without ``bf16_storage`` this is the same function as the plain path up to f32
    rounding (tests/test_encoder_oracle.py pins that on CPU); with it, the rounding points are the folded kernels'.

    ``residual_f32=True`` (with ``bf16_storage``): the HIP path's opt-in form in which the residual stream is NOT rounded between
    layers -- every LayerNorm adds the (rounded) GEMM output to the unrounded previous LayerNorm output; GEMM inputs are rounded.

    ``device``: where torch evaluates these same fp32 expressions.  "cpu" is the oracle proper (and the timed CPU baseline);
    "cuda" runs the identical plain-torch fp32 graph on the GPU (rocBLAS fp32 GEMMs, no bf16, none of this repo's kernels) so
    that end-to-end parity legs can afford thousands of chunks -- tests/test_c2_gpu.py pins it against the CPU evaluation
    (<= 2e-5 relative), and ``weights`` may then be a dict of tensors already on that device (see :func:`to_device`)."""
    def rb(t):
        return t.to(torch.bfloat16).to(dtype) if bf16_storage else t
    W = to_device(weights, device, dtype)
    W0 = W                                    # unrounded masters: the folded forms scale these, then round
    if bf16_storage:
        W = {k: (rb(v) if (v.ndim == 2) else v) for k, v in W.items()}
    ids_t = torch.from_numpy(np.asarray(ids, dtype=np.int64)).to(device)
    B, L = ids_t.shape
    H, nh = cfg.hidden_size, cfg.num_heads
    dh = H // nh
    mask = ids_t.ne(cfg.pad_token_id)
    with torch.no_grad():
        x = W["embeddings.word_embeddings.weight"][ids_t] + W["embeddings.token_type_embeddings.weight"][0]
        x = x + W["embeddings.position_embeddings.weight"][position_ids(ids_t, cfg.pad_token_id)]
        x = rb(_ln(x, W["embeddings.LayerNorm.weight"], W["embeddings.LayerNorm.bias"], cfg.layer_norm_eps))
        neg = torch.finfo(dtype).min
        key_bias = torch.where(mask, 0.0, neg).to(dtype)[:, None, None, :]          # additive, on keys only
        def attention(q, k, v):
            s = (q @ k.transpose(-1, -2)) * (dh ** -0.5) + key_bias
            if bf16_storage:   # the kernel normalises AFTER the P.V product: P = exp(s - max) is what gets rounded
                e = torch.exp(s - s.max(-1, keepdim=True).values)
                ctx = (rb(e) @ v) / e.sum(-1, keepdim=True)
            else:
                ctx = torch.softmax(s, dim=-1) @ v
            return rb(ctx.transpose(1, 2).reshape(B, L, H))

        def row_stats(r):      # (rstd, nmr = -mu rstd) of rows as stored
            mu = r.mean(-1, keepdim=True)
            rstd = 1.0 / torch.sqrt(((r - mu) ** 2).mean(-1, keepdim=True) + cfg.layer_norm_eps)
            return rstd, -mu * rstd

        def lin_lnin(r, st, wname, bname, gain, beta):
            """LayerNorm(r) @ w^T + b as rstd (r @ (w gain)^T) + nmr colsum + (b + w @ beta): gain-scaled weights rounded like weights."""
            ws = rb(W0[wname] * gain[None, :])
            return st[0] * (r @ ws.T) + st[1] * ws.sum(1) + (W0[bname] + W0[wname] @ beta)
        r2 = st2 = None
        for i in (range(cfg.num_layers) if ln_fold else ()):
            p = f"encoder.layer.{i}."
            if i == 0:
                qkv = [rb(x @ W[p + f"attention.self.{n}.weight"].T + W[p + f"attention.self.{n}.bias"]) for n in ("query", "key", "value")]
            else:
                pp = f"encoder.layer.{i - 1}."
                g2, b2 = W[pp + "output.LayerNorm.weight"], W[pp + "output.LayerNorm.bias"]
                qkv = [rb(lin_lnin(r2, st2, p + f"attention.self.{n}.weight", p + f"attention.self.{n}.bias", g2, b2)) for n in ("query", "key", "value")]
            ctx = attention(*(t.view(B, L, nh, dh).transpose(1, 2) for t in qkv))
            o = ctx @ W[p + "attention.output.dense.weight"].T
            if i == 0:
                r1 = rb(o + W[p + "attention.output.dense.bias"] + x)
            else:
                r1 = rb((r2 * st2[0] + st2[1]) * g2 + (o + (W[p + "attention.output.dense.bias"] + b2)))
            st1 = row_stats(r1)
            g1, b1 = W[p + "attention.output.LayerNorm.weight"], W[p + "attention.output.LayerNorm.bias"]
            h = lin_lnin(r1, st1, p + "intermediate.dense.weight", p + "intermediate.dense.bias", g1, b1)
            h = rb(h * 0.5 * (1.0 + torch.erf(h / math.sqrt(2.0))))
            r2 = rb((r1 * st1[0] + st1[1]) * g1 + (h @ W[p + "output.dense.weight"].T + (W[p + "output.dense.bias"] + b1)))
            st2 = row_stats(r2)
            if i == cfg.num_layers - 1:
                x = rb((r2 * st2[0] + st2[1]) * W[p + "output.LayerNorm.weight"] + W[p + "output.LayerNorm.bias"])
        for i in (() if ln_fold else range(cfg.num_layers)):
            p = f"encoder.layer.{i}."

            def lin(t, name):
                return t @ W[p + name + ".weight"].T + W[p + name + ".bias"]
            q = rb(lin(x, "attention.self.query")).view(B, L, nh, dh).transpose(1, 2)
            k = rb(lin(x, "attention.self.key")).view(B, L, nh, dh).transpose(1, 2)
            v = rb(lin(x, "attention.self.value")).view(B, L, nh, dh).transpose(1, 2)
            ctx = attention(q, k, v)
            if residual_f32:     # xr: the unrounded stream; x = rb(xr) is what the GEMMs read
                if i == 0:
                    xr = x
                xr = _ln(rb(lin(ctx, "attention.output.dense")) + xr, W[p + "attention.output.LayerNorm.weight"],
                         W[p + "attention.output.LayerNorm.bias"], cfg.layer_norm_eps)
                h = lin(rb(xr), "intermediate.dense")
                h = rb(h * 0.5 * (1.0 + torch.erf(h / math.sqrt(2.0))))
                xr = _ln(rb(lin(h, "output.dense")) + xr, W[p + "output.LayerNorm.weight"], W[p + "output.LayerNorm.bias"], cfg.layer_norm_eps)
                x = rb(xr)
                continue
            x = rb(_ln(rb(lin(ctx, "attention.output.dense") + x), W[p + "attention.output.LayerNorm.weight"],
                       W[p + "attention.output.LayerNorm.bias"], cfg.layer_norm_eps))
            h = lin(x, "intermediate.dense")
            h = rb(h * 0.5 * (1.0 + torch.erf(h / math.sqrt(2.0))))
            x = rb(_ln(rb(lin(h, "output.dense") + x), W[p + "output.LayerNorm.weight"], W[p + "output.LayerNorm.bias"],
                       cfg.layer_norm_eps))
        m = mask.to(dtype)
        sent = (x * m[..., None]).sum(1) / m.sum(-1)[..., None]
    sent, x = sent.float().cpu(), x.float().cpu()
    return (sent.numpy(), x.numpy()) if return_tokens else sent.numpy()


def to_device(weights: dict, device: str = "cpu", dtype: torch.dtype = torch.float32) -> dict:
    """numpy (or tensor) weights -> tensors of ``dtype`` on ``device``; tensors already there pass through (so a caller
    that evaluates many batches converts once)."""
    out = {}
    for k, v in weights.items():
        t = v if torch.is_tensor(v) else torch.from_numpy(np.ascontiguousarray(v))
        out[k] = t.to(device=device, dtype=dtype)
    return out


def synthetic_ids(cfg: EncoderConfig, lengths, seed: int, enc_only_id: int = 5, pad_to: int | None = None) -> np.ndarray:
    """Token ids shaped like UniXcoder.tokenize output (unixcoder_provider.py:108-122): [<s>=0, <encoder-only>, </s>=2]
    + body + [</s>=2], body uniform over the vocabulary minus the specials; right-padded with the pad id."""
    rng = np.random.default_rng(seed)
    lengths = [int(v) for v in lengths]
    L = pad_to or max(lengths)
    out = np.full((len(lengths), L), cfg.pad_token_id, dtype=np.int64)
    for r, n in enumerate(lengths):
        n = max(4, min(n, L))
        body = rng.integers(3, cfg.vocab_size, size=n - 4)
        out[r, :n] = np.concatenate([[0, enc_only_id, 2], body, [2]])
    return out
