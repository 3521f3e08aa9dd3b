"""HIP UniXcoder encoder driver.

Host-side mirror of ``UniXcoder`` + ``embed_batch_sync`` (``src/lattice/providers/unixcoder_provider.py:52-215``):
tokenise -> ids -> 12-layer post-LN encoder -> masked mean pool -> python floats.  Every tensor op of the forward
is a hand-written HIP kernel reached through the C ABI (``crh_embed_ln``, ``crh_gemm_bf16_bias``,
``crh_attn_fwd_varlen``, ``crh_gemm_bf16_bias_res_ln``, ``crh_masked_mean_pool``); PyTorch only owns the device
buffers and the stream.  Differences from the reference, all deliberate:

* ragged batches are right-padded with the pad id instead of raising (quirk Q1) -- padding is invisible to the
  real tokens (position ids skip pads, pad keys are masked, the pool is masked);
* attention is bidirectional with key masking (what the reference computed under transformers 4.x, quirk Q2);
* texts are length-bucketed so short chunks do not pay for 512-token padding.
"""

from __future__ import annotations

import json
import math
import os
import time
import re
import threading
import zlib
from dataclasses import dataclass

import numpy as np

from . import ffi


def _yield_gil() -> None:
    """Let a thread that is waiting for the interpreter lock have it now (``time.sleep(0)`` releases the lock and yields the CPU)."""
    time.sleep(0)


@dataclass(frozen=True)
class EncoderConfig:
    vocab_size: int = 51416
    hidden_size: int = 768
    num_layers: int = 12
    num_heads: int = 12
    intermediate_size: int = 3072
    max_position_embeddings: int = 1026
    type_vocab_size: int = 10
    layer_norm_eps: float = 1e-5
    pad_token_id: int = 1
    # LayerNorm folded into the GEMMs around it (csrc/crh_encoder.hip, "LayerNorm folded ..."): the residual stream stays
    # un-normalised between kernels, no [T, 768] pass just to normalise.  Built, bit-stable across batch sizes, and MEASURED
    # 1.9 % SLOWER than the two LayerNorm kernels per layer at 65 k tokens (profiles/r05_ln_fold_ab.md: the producer GEMMs'
    # epilogue work -- the residual's LayerNorm and the row statistics -- runs with the matrix pipe idle and costs what the two
    # memory-bound passes cost), so it is OPT-IN: its results sit slightly closer to the fp32 forward (one rounding of a
    # normalised activation less per LayerNorm: tests/test_encoder_oracle.py), which is what one may want it for.
    ln_fold: bool = os.environ.get("CODERAG_HIP_LN_FOLD", "0") == "1"
    # The residual stream in f32 (opt-in fidelity lever, round 5): what each layer adds its output to is not rounded to bf16
    # between layers; the GEMMs are unchanged (bf16 rows in, bf16 out), only the two LayerNorm kernels read and write an f32 copy
    # beside the bf16 one (12 bytes per element instead of 6: measured cost and gain in bench.py's c2 record / DESIGN.md section 4c).
    residual_f32: bool = os.environ.get("CODERAG_HIP_RESIDUAL_F32", "0") == "1"

    @classmethod
    def from_hf_json(cls, path: str) -> "EncoderConfig":
        c = json.load(open(path))
        return cls(vocab_size=c["vocab_size"], hidden_size=c["hidden_size"], num_layers=c["num_hidden_layers"],
                   num_heads=c["num_attention_heads"], intermediate_size=c["intermediate_size"],
                   max_position_embeddings=c["max_position_embeddings"], type_vocab_size=c.get("type_vocab_size", 1),
                   layer_norm_eps=c.get("layer_norm_eps", 1e-5), pad_token_id=c.get("pad_token_id", 1))


def synthetic_weights(cfg: EncoderConfig, seed: int, init: str = "sharp") -> dict[str, np.ndarray]:
    """Seeded stand-in weights under HF ``RobertaModel`` state-dict names (no checkpoint exists offline).  The numpy
    Generator stream is machine-independent, so every box regenerates the same tensors.  ``init="hf"``: the statistics of
    HF's ``_init_weights`` (N(0, 0.02^2) matrices and tables, zero biases, unit LayerNorm) instead of the sharp O(1) ones;
    ``init="hf_ln"``: between the two -- HF-init matrices with the sharp fixture's biases and LayerNorm gains / biases (a trained
    checkpoint's LayerNorm parameters are not 1 / 0: bench.py's fidelity leg reports all three)."""
    rng = np.random.default_rng(seed)
    H, F = cfg.hidden_size, cfg.intermediate_size

    hf = init in ("hf", "hf_ln")          # matrices and tables N(0, 0.02^2)
    hf_vec = init == "hf"                 # ... and zero biases / unit LayerNorm; "hf_ln" keeps the sharp fixture's biases and LayerNorm gains
    if init not in ("sharp", "hf", "hf_ln"):
        raise ValueError(f"unknown init {init!r}")

    def mat(n, k, std):
        std = 0.02 if hf else std
        return rng.standard_normal((n, k), dtype=np.float32) * np.float32(std)

    def vec(n, std, mean=0.0):
        if hf_vec:   # (the draw is still made, so all flavours consume the generator identically)
            return rng.standard_normal(n, dtype=np.float32) * np.float32(0.0) + np.float32(mean)
        return rng.standard_normal(n, dtype=np.float32) * np.float32(std) + np.float32(mean)
    w = {
        "embeddings.word_embeddings.weight": mat(cfg.vocab_size, H, 0.5),
        "embeddings.position_embeddings.weight": mat(cfg.max_position_embeddings, H, 0.3),
        "embeddings.token_type_embeddings.weight": mat(cfg.type_vocab_size, H, 0.1),
        "embeddings.LayerNorm.weight": vec(H, 0.1, 1.0),
        "embeddings.LayerNorm.bias": vec(H, 0.05),
    }
    w["embeddings.word_embeddings.weight"][cfg.pad_token_id] = 0.0
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


_PIECES = re.compile(r"[A-Za-z_]+|\d+|\s+|[^\sA-Za-z_\d]")


class HashTokenizer:
    """Deterministic stand-in for the byte-level BPE tokenizer when no ``vocab.json``/``merges.txt`` is available
    (benchmarks, tests, synthetic weights): splits identifiers / numbers / whitespace / punctuation and hashes every
    piece into the vocabulary above the special ids.  NOT UniXcoder's vocabulary -- only the id *shape* is the same."""

    cls_id, pad_id, sep_id = 0, 1, 2

    def __init__(self, vocab_size: int, enc_only_id: int = 5):
        self.vocab_size, self.enc_only_id = vocab_size, enc_only_id

    def encode_body(self, text: str) -> list[int]:
        span = self.vocab_size - 16
        return [16 + zlib.crc32(p.encode("utf-8")) % span for p in _PIECES.findall(text)]


class BpeTokenizer:
    """The real thing, from LOCAL files only (``RobertaTokenizer(vocab_file, merges_file)``; never a hub name)."""

    def __init__(self, directory: str):
        from transformers import RobertaTokenizer
        self._tok = RobertaTokenizer(os.path.join(directory, "vocab.json"), os.path.join(directory, "merges.txt"))
        self.cls_id, self.sep_id, self.pad_id = self._tok.cls_token_id, self._tok.sep_token_id, self._tok.pad_token_id
        eid = self._tok.convert_tokens_to_ids("<encoder-only>")          # from the vocabulary, never hard-coded
        if eid is None or eid == self._tok.unk_token_id:
            raise ValueError("vocab.json has no <encoder-only> token: not a UniXcoder vocabulary")
        self.enc_only_id = eid

    def encode_body(self, text: str) -> list[int]:
        return self._tok.convert_tokens_to_ids(self._tok.tokenize(text))


def wrap_encoder_only(tok, text: str, max_length: int = 512) -> list[int]:
    """``UniXcoder.tokenize(mode="<encoder-only>")`` (unixcoder_provider.py:105-122): body truncated to
    ``max_length - 4`` pieces, wrapped as [<s>, <encoder-only>, </s>] + body + [</s>]."""
    assert max_length < 1024
    body = tok.encode_body(text)[: max_length - 4]
    return [tok.cls_id, tok.enc_only_id, tok.sep_id] + body + [tok.sep_id]


def flops_per_chunk(L: int, cfg: EncoderConfig = EncoderConfig()) -> float:
    """Algorithmic FLOPs of one chunk of L real tokens: GEMMs + attention contractions."""
    H, F, n = cfg.hidden_size, cfg.intermediate_size, cfg.num_layers
    return n * (2.0 * L * (4 * H * H + 2 * H * F) + 4.0 * L * L * H)


class HipUniXcoder:
    """Weights resident on one GPU + the kernel driver."""

    def __init__(self, weights: dict, cfg: EncoderConfig, tokenizer, device: int = 0):
        import torch
        if cfg.hidden_size != 768 or cfg.num_heads * 64 != cfg.hidden_size:
            raise ValueError("the HIP encoder kernels are built for hidden 768 = 12 heads x 64")
        ffi.lib()
        self.cfg, self.tok, self.device = cfg, tokenizer, torch.device("cuda", device)
        self._torch = torch

        def dev(name, dtype):
            t = weights[name]
            t = torch.from_numpy(np.ascontiguousarray(t)) if isinstance(t, np.ndarray) else t
            return t.to(device=self.device, dtype=dtype).contiguous()
        bf, f32 = torch.bfloat16, torch.float32
        self.word = dev("embeddings.word_embeddings.weight", bf)
        self.pos = dev("embeddings.position_embeddings.weight", bf)
        self.type0 = dev("embeddings.token_type_embeddings.weight", bf)[0].contiguous()
        self.emb_g, self.emb_b = dev("embeddings.LayerNorm.weight", f32), dev("embeddings.LayerNorm.bias", f32)
        self.layers = []

        def host(name):      # f32 master on the host (the folding below is done once, in f64, from the unrounded weights)
            t = weights[name]
            t = torch.from_numpy(np.ascontiguousarray(t)) if isinstance(t, np.ndarray) else t
            return t.detach().to(device="cpu", dtype=torch.float64)

        def fold(w64, b64, gain64, beta64):
            """LayerNorm(x) @ w^T + b = rstd (x @ (w * gain)^T - mu * colsum) + (b + w @ beta): the gain-scaled weights (bf16),
            the column sums OF THE ROUNDED VALUES (what the matrix pipe actually adds up) and the folded bias."""
            ws = (w64 * gain64[None, :]).to(torch.float32).to(bf)
            colsum = ws.to(torch.float64).sum(1).to(torch.float32)
            bias = (b64 + w64 @ beta64).to(torch.float32)
            return ws.to(self.device).contiguous(), colsum.to(self.device).contiguous(), bias.to(self.device).contiguous()
        for i in range(cfg.num_layers):
            p = f"encoder.layer.{i}."
            qkv_w = torch.cat([dev(p + f"attention.self.{n}.weight", bf) for n in ("query", "key", "value")], 0).contiguous()
            qkv_b = torch.cat([dev(p + f"attention.self.{n}.bias", f32) for n in ("query", "key", "value")], 0).contiguous()
            ly = dict(
                qkv_w=qkv_w, qkv_b=qkv_b,
                o_w=dev(p + "attention.output.dense.weight", bf), o_b=dev(p + "attention.output.dense.bias", f32),
                ln1_g=dev(p + "attention.output.LayerNorm.weight", f32), ln1_b=dev(p + "attention.output.LayerNorm.bias", f32),
                f1_w=dev(p + "intermediate.dense.weight", bf), f1_b=dev(p + "intermediate.dense.bias", f32),
                f2_w=dev(p + "output.dense.weight", bf), f2_b=dev(p + "output.dense.bias", f32),
                ln2_g=dev(p + "output.LayerNorm.weight", f32), ln2_b=dev(p + "output.LayerNorm.bias", f32))
            if cfg.ln_fold and cfg.residual_f32:
                raise ValueError("ln_fold and residual_f32 are two forms of the LayerNorm step: choose one")
            if cfg.ln_fold:
                g1, b1 = host(p + "attention.output.LayerNorm.weight"), host(p + "attention.output.LayerNorm.bias")
                ly["f1_ws"], ly["f1_c"], ly["f1_bf"] = fold(host(p + "intermediate.dense.weight"), host(p + "intermediate.dense.bias"), g1, b1)
                ly["f2_bf"] = (host(p + "output.dense.bias") + b1).to(torch.float32).to(self.device).contiguous()      # FFN2's bias + LayerNorm-1's beta
                del ly["f1_w"]                   # (the unscaled FFN1 weights are not used by the folded forward)
                if i > 0:                        # layer 0 reads the embedding LayerNorm's output, which is materialised
                    q = f"encoder.layer.{i - 1}."
                    g2, b2 = host(q + "output.LayerNorm.weight"), host(q + "output.LayerNorm.bias")
                    wq = torch.cat([host(p + f"attention.self.{n}.weight") for n in ("query", "key", "value")], 0)
                    bq = torch.cat([host(p + f"attention.self.{n}.bias") for n in ("query", "key", "value")], 0)
                    ly["qkv_ws"], ly["qkv_c"], ly["qkv_bf"] = fold(wq, bq, g2, b2)
                    ly["o_bf"] = (host(p + "attention.output.dense.bias") + b2).to(torch.float32).to(self.device).contiguous()
                    del ly["qkv_w"]
            self.layers.append(ly)
        # raw addresses, taken once: a forward is 62 launches with ~10 pointer arguments each, and on the one-query path the
        # host's enqueue time is as long as the GPU's chain of kernels
        # models are shared process-wide (load_unixcoder) while the pinned staging slots below are per model: one submission at a time
        self._submit_lock = threading.Lock()
        self._emb_ptrs = tuple(int(t.data_ptr()) for t in (self.word, self.pos, self.type0, self.emb_g, self.emb_b))
        self._layer_ptrs = [{k: int(v.data_ptr()) for k, v in ly.items()} for ly in self.layers]

    # ------------------------------------------------------------------ kernels
    def forward_ids(self, ids):
        """ids: int32 CUDA tensor [B, L], L % 16 == 0, padded with the pad id.  Returns f32 [B, 768] sentence embeddings."""
        torch, L_ = self._torch, ffi.lib()
        ffi.use_device(self.device.index)      # (worker threads start on device 0)
        B, L = ids.shape
        cfg, H, F = self.cfg, self.cfg.hidden_size, self.cfg.intermediate_size
        T = B * L
        st = torch.cuda.current_stream(self.device).cuda_stream
        bf = torch.bfloat16
        x = torch.empty((T, H), dtype=bf, device=self.device)
        x1 = torch.empty((T, H), dtype=bf, device=self.device)
        qkv = torch.empty((T, 3 * H), dtype=bf, device=self.device)
        ctx = torch.empty((T, H), dtype=bf, device=self.device)
        hid = torch.empty((T, F), dtype=bf, device=self.device)
        kmask = torch.empty((B, (L + 63) // 64), dtype=torch.int64, device=self.device)
        sent = torch.empty((B, H), dtype=torch.float32, device=self.device)
        px, px1, pqkv, pctx, phid, pkm, psent = (int(t.data_ptr()) for t in (x, x1, qkv, ctx, hid, kmask, sent))
        eps, check = cfg.layer_norm_eps, ffi.check
        gemm, gemm_ln, attn = L_.crh_gemm_bf16_bias, L_.crh_gemm_bf16_bias_res_ln, L_.crh_attn_fwd_varlen
        check(L_.crh_embed_ln(int(ids.data_ptr()), *self._emb_ptrs, eps, cfg.pad_token_id, px, pkm, B, L, H, st))
        if cfg.ln_fold or cfg.residual_f32:
            layers = self._layers_folded if cfg.ln_fold else self._layers_res32
            layers(px, px1, pqkv, pctx, phid, T, st, lambda: check(attn(pqkv, pkm, pctx, B, L, cfg.num_heads, st)), x0=x)
            check(L_.crh_masked_mean_pool(px, pkm, psent, B, L, H, st))
            return sent
        for ly in self._layer_ptrs:
            check(gemm(px, ly["qkv_w"], ly["qkv_b"], pqkv, T, 3 * H, H, 0, st))
            check(attn(pqkv, pkm, pctx, B, L, cfg.num_heads, st))
            check(gemm_ln(pctx, ly["o_w"], ly["o_b"], px, ly["ln1_g"], ly["ln1_b"], eps, px1, T, H, H, st))
            check(gemm(px1, ly["f1_w"], ly["f1_b"], phid, T, F, H, 1, st))
            check(gemm_ln(phid, ly["f2_w"], ly["f2_b"], px1, ly["ln2_g"], ly["ln2_b"], eps, px, T, H, F, st))
        check(L_.crh_masked_mean_pool(px, pkm, psent, B, L, H, st))
        return sent

    def _layers_res32(self, px, px1, pqkv, pctx, phid, T, st, attention, x0=None) -> None:
        """The twelve layers with the residual stream in f32 (EncoderConfig.residual_f32): the same GEMMs and attention on bf16
        rows; each LayerNorm adds the GEMM's bf16 output to the F32 residual, writes bf16 for the next GEMM and f32 for the next
        residual (csrc/crh_encoder.hip, k_layernorm768_res32).  ``px`` holds the embedding LayerNorm's output on entry (bf16: the
        stream starts from it) and the last LayerNorm's bf16 output on exit."""
        torch, L_ = self._torch, ffi.lib()
        cfg, H, F = self.cfg, self.cfg.hidden_size, self.cfg.intermediate_size
        r32 = x0.to(torch.float32)          # the stream starts from the embedding LayerNorm's bf16 output
        pr32 = int(r32.data_ptr())
        eps, check = cfg.layer_norm_eps, ffi.check
        gemm, gemm_ln32 = L_.crh_gemm_bf16_bias, L_.crh_gemm_bf16_bias_res32_ln
        for ly in self._layer_ptrs:
            _yield_gil()
            check(gemm(px, ly["qkv_w"], ly["qkv_b"], pqkv, T, 3 * H, H, 0, st))
            attention()
            check(gemm_ln32(pctx, ly["o_w"], ly["o_b"], pr32, ly["ln1_g"], ly["ln1_b"], eps, px1, T, H, H, st))
            check(gemm(px1, ly["f1_w"], ly["f1_b"], phid, T, F, H, 1, st))
            check(gemm_ln32(phid, ly["f2_w"], ly["f2_b"], pr32, ly["ln2_g"], ly["ln2_b"], eps, px, T, H, F, st))

    def _layers_folded(self, px, px1, pqkv, pctx, phid, T, st, attention, x0=None) -> None:
        """The twelve layers with the LayerNorms folded into the GEMMs around them (csrc/crh_encoder.hip, "LayerNorm folded into
        the GEMMs around it"; reference arithmetic: modeling_roberta.py:329-340, 387-398).  ``px`` holds the embedding LayerNorm's
        output on entry and the LAST LayerNorm's output on exit; in between the residual stream is un-normalised rows + per-row
        (rstd, -mu rstd): per layer 4 GEMMs (two of them followed by the tiny statistics kernel) + attention."""
        torch, L_ = self._torch, ffi.lib()
        cfg, H, F = self.cfg, self.cfg.hidden_size, self.cfg.intermediate_size
        st1 = torch.empty((T, 2), dtype=torch.float32, device=self.device)
        st2 = torch.empty((T, 2), dtype=torch.float32, device=self.device)
        part = torch.empty((T, H // 32, 2), dtype=torch.float32, device=self.device)
        pst1, pst2, ppart = int(st1.data_ptr()), int(st2.data_ptr()), int(part.data_ptr())
        eps, check = cfg.layer_norm_eps, ffi.check
        gemm, lnin, res = L_.crh_gemm_bf16_bias, L_.crh_gemm_bf16_lnin, L_.crh_gemm_bf16_res_lnstats
        prev = None
        for ly in self._layer_ptrs:
            _yield_gil()
            if prev is None:
                check(gemm(px, ly["qkv_w"], ly["qkv_b"], pqkv, T, 3 * H, H, 0, st))
            else:
                check(lnin(px, pst2, ly["qkv_ws"], ly["qkv_c"], ly["qkv_bf"], pqkv, T, 3 * H, H, 0, st))
            attention()
            if prev is None:        # residual = the embedding LayerNorm's output, as it is
                check(res(pctx, ly["o_w"], ly["o_b"], px, None, None, eps, px1, ppart, pst1, T, H, H, st))
            else:                   # residual = LayerNorm-2 of the previous layer, worked out from its un-normalised rows
                check(res(pctx, ly["o_w"], ly["o_bf"], px, pst2, prev["ln2_g"], eps, px1, ppart, pst1, T, H, H, st))
            check(lnin(px1, pst1, ly["f1_ws"], ly["f1_c"], ly["f1_bf"], phid, T, F, H, 1, st))
            check(res(phid, ly["f2_w"], ly["f2_bf"], px1, pst1, ly["ln1_g"], eps, px, ppart, pst2, T, H, F, st))
            prev = ly
        check(L_.crh_layernorm_apply(px, pst2, prev["ln2_g"], prev["ln2_b"], px, T, H, st))
        # (st1 / st2 / part go back to torch's caching allocator here while the launches above may still be queued: the allocator
        # hands a block out again only to work ordered behind them on this stream, like every other buffer of a forward)

    def forward_packed(self, ids, row_off, Lmax: int, verify: bool = False):
        """The forward on a batch WITHOUT padding: ``ids`` int32 CUDA tensor [T] (the rows' tokens back to back), ``row_off``
        int32 CUDA tensor [B + 1] (row b = ids[row_off[b]:row_off[b+1]]), ``Lmax`` a multiple of 16 >= every row's length.
        Returns f32 [B, 768].  GEMMs and LayerNorms run on the T real tokens; attention, the embedding gather and the pool
        take the row offsets (``crh_*_packed``).  Same arithmetic per token as :meth:`forward_ids`.  The kernels clamp every row to
        the T tokens of the buffers and a device-side check of ``row_off`` reports a bad array as ``NativeError(E_INVALID)`` at the
        next packed call -- or right here with ``verify=True`` (which waits for the stream)."""
        torch, L_ = self._torch, ffi.lib()
        ffi.use_device(self.device.index)
        B, T = int(row_off.shape[0]) - 1, int(ids.shape[0])
        cfg, H, F = self.cfg, self.cfg.hidden_size, self.cfg.intermediate_size
        st = torch.cuda.current_stream(self.device).cuda_stream
        bf = torch.bfloat16
        x = torch.empty((T, H), dtype=bf, device=self.device)
        x1 = torch.empty((T, H), dtype=bf, device=self.device)
        qkv = torch.empty((T, 3 * H), dtype=bf, device=self.device)
        ctx = torch.empty((T, H), dtype=bf, device=self.device)
        hid = torch.empty((T, F), dtype=bf, device=self.device)
        kmask = torch.empty((B, (Lmax + 63) // 64), dtype=torch.int64, device=self.device)
        sent = torch.empty((B, H), dtype=torch.float32, device=self.device)
        px, px1, pqkv, pctx, phid, pkm, psent, poff = (int(t.data_ptr()) for t in (x, x1, qkv, ctx, hid, kmask, sent, row_off))
        eps, check = cfg.layer_norm_eps, ffi.check
        gemm, gemm_ln = L_.crh_gemm_bf16_bias, L_.crh_gemm_bf16_bias_res_ln
        check(L_.crh_embed_ln_packed(int(ids.data_ptr()), poff, *self._emb_ptrs, eps, cfg.pad_token_id, px, pkm, B, T, Lmax, H, st))
        if cfg.ln_fold or cfg.residual_f32:
            layers = self._layers_folded if cfg.ln_fold else self._layers_res32
            layers(px, px1, pqkv, pctx, phid, T, st, lambda: check(L_.crh_attn_fwd_packed(pqkv, poff, pkm, pctx, B, T, Lmax, cfg.num_heads, st)), x0=x)
        for ly in (() if (cfg.ln_fold or cfg.residual_f32) else self._layer_ptrs):
            # (a launch loop re-takes the interpreter lock microseconds after every ctypes call: a thread that waits for it -- the
            # store's worker answering a query beside this indexing run -- would otherwise get it only at the 5 ms switch
            # interval, per hop; yielding once per layer costs a microsecond and lets it in within a layer's ~1 ms)
            _yield_gil()
            check(gemm(px, ly["qkv_w"], ly["qkv_b"], pqkv, T, 3 * H, H, 0, st))
            check(L_.crh_attn_fwd_packed(pqkv, poff, pkm, pctx, B, T, Lmax, cfg.num_heads, st))
            check(gemm_ln(pctx, ly["o_w"], ly["o_b"], px, ly["ln1_g"], ly["ln1_b"], eps, px1, T, H, H, st))
            check(gemm(px1, ly["f1_w"], ly["f1_b"], phid, T, F, H, 1, st))
            check(gemm_ln(phid, ly["f2_w"], ly["f2_b"], px1, ly["ln2_g"], ly["ln2_b"], eps, px, T, H, F, st))
        check(L_.crh_masked_mean_pool_packed(px, poff, pkm, psent, B, T, Lmax, H, st))
        if verify:      # the sync point of a forward: the verdict of the device-side check of row_off (crh_encoder_finish)
            check(L_.crh_encoder_finish(st))
        return sent

    def pack_rows(self, id_rows, rows):
        """Host side of a packed batch: (ids int32 [T], row_off int32 [B + 1], Lmax) for the id lists ``id_rows[i], i in rows``."""
        lens = np.fromiter((len(id_rows[i]) for i in rows), dtype=np.int64, count=len(rows))
        off = np.zeros(len(rows) + 1, dtype=np.int32)
        np.cumsum(lens, out=off[1:])
        flat = np.empty(int(off[-1]), dtype=np.int32)
        for r, i in enumerate(rows):
            flat[off[r]:off[r + 1]] = id_rows[i]
        return flat, off, (int(lens.max()) + 15) // 16 * 16

    # ------------------------------------------------------------------ batching
    def plan_batches(self, lengths, max_tokens: int = 65536, max_rows: int = 1024, packed: bool = False):
        """Length-bucketed batches: rows sorted by length, each batch padded to its longest row rounded up to 16 (the
        query-tile height of the attention kernel): ~4 % padded tokens on a mean-200 mix, against 16 % at granularity 64.
        ``packed=True``: the batches of :meth:`forward_packed` -- no padding at all, so a batch is filled to ``max_tokens`` REAL
        tokens (the second element is still the longest row rounded up to 16: it sizes the attention launch)."""
        order = np.argsort(np.asarray(lengths), kind="stable")
        batches, cur, cur_L, cur_T = [], [], 0, 0
        for i in order:
            n = int(lengths[i])
            L = (n + 15) // 16 * 16
            newL = max(cur_L, L)
            full = (cur_T + n > max_tokens) if packed else (newL * (len(cur) + 1) > max_tokens)
            if cur and (full or len(cur) >= max_rows):
                batches.append((cur, cur_L))
                cur, newL, cur_T = [], L, 0
            cur.append(int(i))
            cur_L = newL
            cur_T += n
        if cur:
            batches.append((cur, cur_L))
        return batches

    def _check_ids(self, ids: np.ndarray, lens=None) -> None:
        """The embedding gather trusts its indices: an id outside the word-embedding table (a vocabulary that does not
        belong to the checkpoint) must stop here, not fault on the device."""
        if ids.size == 0:
            return
        if lens is not None:
            ids = np.where(np.arange(ids.shape[1])[None, :] < np.asarray(lens)[:, None], ids, 0)
        lo, hi = int(ids.min()), int(ids.max())
        if lo < 0 or hi >= self.cfg.vocab_size:
            raise ValueError(f"token id {hi if hi >= self.cfg.vocab_size else lo} outside the embedding table (vocab_size {self.cfg.vocab_size})")

    def _pinned(self, slot: int, name: str, n: int, dtype):
        """A cached pinned host buffer of at least n elements (pinning is slow; calls pack into the same few buffers).  Two
        slots alternate, each guarded by the event recorded behind the copies that last read it."""
        torch = self._torch
        cache = self.__dict__.setdefault("_pin_cache", {})
        buf = cache.get((slot, name))
        if buf is None or buf.numel() < n or buf.dtype != dtype:
            buf = torch.empty((max(n, 1 << 16),), dtype=dtype, pin_memory=True)
            cache[(slot, name)] = buf
        return buf[:n]

    def _embed_packed(self, batches, lens, fill, n: int):
        """Run packed batches.  ``batches``: [(rows, Lmax)] from plan_batches(packed=True); ``lens[i]``: tokens of input row i;
        ``fill(rows, row_lens, dst, starts)``: write the ids of a batch's rows back to back into the int32 numpy view ``dst``
        (row r begins at ``starts[r]``).  Everything the device needs from the host -- the
        ids of ALL batches back to back, their row offsets, the scatter order -- goes up in THREE asynchronous copies from
        pinned memory before the first kernel, so the host never waits on the stream while it enqueues (a pageable copy per
        batch kept the host in lockstep with the GPU, and nothing else could overlap with it).  Returns f32 CUDA [n, 768]."""
        with self._submit_lock:      # (two providers / a query thread beside an indexing thread share this model's pinned slots)
            return self._embed_packed_locked(batches, lens, fill, n)

    def _embed_packed_locked(self, batches, lens, fill, n: int):
        torch = self._torch
        ffi.use_device(self.device.index)
        total = int(sum(int(lens[i]) for rows, _ in batches for i in rows))
        nrows = sum(len(rows) for rows, _ in batches)
        st = self.__dict__.setdefault("_pin_state", {"slot": 0, "events": [None, None]})
        slot = st["slot"] = st["slot"] ^ 1
        if st["events"][slot] is not None:
            st["events"][slot].synchronize()          # the copies that last read this slot's buffers have run
        p_ids = self._pinned(slot, "ids", total, torch.int32)
        p_off = self._pinned(slot, "off", nrows + len(batches), torch.int32)
        p_ord = self._pinned(slot, "ord", nrows, torch.int64)
        h_ids, h_off, h_ord = p_ids.numpy(), p_off.numpy(), p_ord.numpy()
        spans, t0, o0, r0 = [], 0, 0, 0
        for rows, Lmax in batches:
            bl = np.asarray(lens)[rows]
            h_off[o0] = 0
            np.cumsum(bl, out=h_off[o0 + 1:o0 + 1 + len(rows)])
            pos = int(h_off[o0 + len(rows)])
            fill(rows, bl, h_ids[t0:t0 + pos], h_off[o0:o0 + len(rows)])
            h_ord[r0:r0 + len(rows)] = rows
            spans.append((t0, pos, o0, len(rows), r0, Lmax))
            t0, o0, r0 = t0 + pos, o0 + len(rows) + 1, r0 + len(rows)
        d_ids = p_ids.to(self.device, non_blocking=True)
        d_off = p_off.to(self.device, non_blocking=True)
        d_ord = p_ord.to(self.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        st["events"][slot] = ev
        if n == 1:                         # the query path: no scatter through a device index
            return self.forward_packed(d_ids, d_off, spans[0][5])
        out = torch.empty((n, self.cfg.hidden_size), dtype=torch.float32, device=self.device)
        for t0, nt, o0, nr, r0, Lmax in spans:
            out[d_ord[r0:r0 + nr]] = self.forward_packed(d_ids[t0:t0 + nt], d_off[o0:o0 + nr + 1], Lmax)
        return out

    def embed_ids(self, id_lists, max_tokens: int = 65536):
        """list of token-id lists (each <= 512) -> f32 CUDA tensor [n, 768] in input order."""
        n = len(id_lists)
        for x in id_lists:
            if len(x) == 0:
                raise ValueError("an empty token list cannot be embedded (the reference's tokenize() always emits 4 specials)")
            if min(x) < 0 or max(x) >= self.cfg.vocab_size:
                raise ValueError(f"token id outside the embedding table (vocab_size {self.cfg.vocab_size})")
            if len(x) > self.cfg.max_position_embeddings - 2:
                raise ValueError(f"{len(x)} tokens exceed the position table ({self.cfg.max_position_embeddings - 2})")
        lens = np.fromiter((len(x) for x in id_lists), dtype=np.int64, count=n)

        def fill(rows, row_lens, dst, starts):
            for r, i in enumerate(rows):
                dst[starts[r]:starts[r] + row_lens[r]] = id_lists[i]
        return self._embed_packed(self.plan_batches(lens, max_tokens, max_rows=4096, packed=True), lens, fill, n)

    def embed_bodies(self, body_ids: np.ndarray, body_lens: np.ndarray, max_length: int = 512, max_tokens: int = 65536):
        """Batch form of tokenize + wrap + embed for a tokenizer that returns an id matrix (``NativeBpeTokenizer``):
        ``body_ids`` int32 [n, >= max_length - 4], ``body_lens`` the untruncated counts.  Rows are wrapped as
        [<s>, <encoder-only>, </s>] + body[: max_length - 4] + [</s>] (unixcoder_provider.py:105-122) while being packed into
        the batch arrays -- no per-token Python work."""
        tok = self.tok
        n = int(body_lens.shape[0])
        blen = np.minimum(body_lens.astype(np.int64), max_length - 4)
        self._check_ids(body_ids[:, : int(blen.max()) if n else 0], blen)
        head = np.asarray([tok.cls_id, tok.enc_only_id, tok.sep_id], dtype=np.int32)
        sep = tok.sep_id

        def fill(rows, row_lens, dst, starts):      # whole batch at once: [specials | body | </s>] rows, then the valid cells in order
            b = row_lens - 4
            width = int(b.max()) + 4
            arr = np.empty((len(rows), width), dtype=np.int32)
            arr[:, :3] = head
            arr[:, 3:width - 1] = body_ids[rows, : width - 4]
            arr[np.arange(len(rows)), 3 + b] = sep
            dst[:] = arr[np.arange(width)[None, :] < row_lens[:, None]]
        lens = blen + 4
        return self._embed_packed(self.plan_batches(lens, max_tokens, max_rows=4096, packed=True), lens, fill, n)

    PIPELINE_CHUNK = 8192      # texts per stage of the text pipeline below

    def embed_texts(self, texts, max_length: int = 512, rows: str = "list"):
        """``embed_batch_sync`` (unixcoder_provider.py:195-215): one 768-vector of python floats per text.  ``rows="numpy"``
        returns the rows as float32 numpy views instead (a list of 768-arrays): turning 768 floats per text into Python
        floats and back into an array at the store costs as much as the GPU forward (measured 0.26 s + 0.5 s per 20 k texts).

        With the native tokenizer large calls run as a three-stage pipeline over chunks of ``PIPELINE_CHUNK`` texts: the C++
        tokenizer (its own threads, GIL released) works on chunk i+1 and the float-list conversion on chunk i-1 while the GPU
        runs chunk i; results come back through a side stream into pinned memory.  (Sequentially the host stages were a third
        of the call: 20 k texts took 1.22 s of which the forward 0.8.)"""
        if not texts:
            return np.zeros((0, self.cfg.hidden_size), np.float32) if rows == "array" else []
        texts = list(texts)

        def shaped(out):          # one [n, 768] float32 array ("array": no per-row objects at all), its rows ("numpy"), or python floats
            return out if rows == "array" else list(out) if rows == "numpy" else out.tolist()
        if not hasattr(self.tok, "encode_bodies"):
            return shaped(self.embed_ids([wrap_encoder_only(self.tok, t, max_length) for t in texts]).cpu().numpy())
        torch = self._torch
        ffi.use_device(self.device.index)
        C = self.PIPELINE_CHUNK
        if len(texts) <= C:
            body_ids, body_lens = self.tok.encode_bodies(texts, max_body=max_length - 4)
            return shaped(self.embed_bodies(body_ids, body_lens, max_length).cpu().numpy())
        from concurrent.futures import ThreadPoolExecutor
        chunks = [texts[i:i + C] for i in range(0, len(texts), C)]
        main = torch.cuda.current_stream(self.device)
        side = torch.cuda.Stream(device=self.device)
        result: list = []

        def drain(item):
            host, done = item
            done.synchronize()
            if rows == "array":
                result.append(host.numpy())
            else:
                result.extend(list(host.numpy()) if rows == "numpy" else host.numpy().tolist())
        with ThreadPoolExecutor(max_workers=1, thread_name_prefix="hip-tokenize") as ex:
            fut = ex.submit(self.tok.encode_bodies, chunks[0], max_length - 4)
            waiting = None
            for k in range(len(chunks)):
                body_ids, body_lens = fut.result()
                if k + 1 < len(chunks):
                    fut = ex.submit(self.tok.encode_bodies, chunks[k + 1], max_length - 4)
                dev = self.embed_bodies(body_ids, body_lens, max_length)          # enqueued on the main stream
                ready = torch.cuda.Event()
                ready.record(main)
                host = torch.empty(dev.shape, dtype=dev.dtype, pin_memory=True)
                with torch.cuda.stream(side):
                    side.wait_event(ready)
                    host.copy_(dev, non_blocking=True)
                    dev.record_stream(side)
                    done = torch.cuda.Event()
                    done.record(side)
                if waiting is not None:
                    drain(waiting)                                                # chunk k-1 -> python floats while the GPU runs chunk k
                waiting = (host, done)
            drain(waiting)
        return np.concatenate(result) if rows == "array" else result


_MODELS: dict = {}


def load_unixcoder(model: str, extra: dict | None = None, device: int | None = None) -> HipUniXcoder:
    """Process-wide singleton per (model, device), like the reference's lru-cached ``get_unixcoder_model``
    (unixcoder_provider.py:157-174).  ``model`` is a LOCAL checkpoint directory, or -- with
    ``extra={"synthetic_weights": seed}`` -- ignored in favour of seeded stand-in weights + the hashing tokenizer."""
    from .settings import get_settings
    extra = extra or {}
    device = get_settings().hip_device if device is None else device
    # the two opt-in forms of the LayerNorm step (EncoderConfig: fidelity levers) can be asked for per provider
    # (``ProviderConfig.extra={"residual_f32": True}``); unset, the environment's / the dataclass's defaults hold
    forms = {k: bool(extra[k]) for k in ("ln_fold", "residual_f32") if k in extra}
    key = (model, device, extra.get("synthetic_weights"), tuple(sorted(forms.items())))
    if key in _MODELS:
        return _MODELS[key]
    if extra.get("synthetic_weights") is not None:
        cfg = EncoderConfig(num_layers=int(extra.get("num_layers", 12)), **forms)
        m = HipUniXcoder(synthetic_weights(cfg, int(extra["synthetic_weights"])), cfg, HashTokenizer(cfg.vocab_size), device)
    elif os.path.isdir(model):
        import torch
        cfg = EncoderConfig.from_hf_json(os.path.join(model, "config.json"))
        if forms:
            import dataclasses
            cfg = dataclasses.replace(cfg, **forms)
        st_path, bin_path = os.path.join(model, "model.safetensors"), os.path.join(model, "pytorch_model.bin")
        if os.path.exists(st_path):
            from safetensors.torch import load_file
            sd = load_file(st_path)
        elif os.path.exists(bin_path):
            sd = torch.load(bin_path, map_location="cpu", weights_only=True)
        else:
            raise FileNotFoundError(f"{model} holds neither model.safetensors nor pytorch_model.bin")
        sd = {k[len("roberta."):] if k.startswith("roberta.") else k: v for k, v in sd.items()}
        # tokenizer: the native byte-level BPE (same ids as the HF one, ~10x its rate; tests/test_tokenizer_native.py) unless
        # CODERAG_TOKENIZER=hf asks for transformers' RobertaTokenizer
        if os.environ.get("CODERAG_TOKENIZER", "native").lower() == "hf":
            tok = BpeTokenizer(model)
        else:
            from .tokenizer_native import NativeBpeTokenizer
            tok = NativeBpeTokenizer(model)
        m = HipUniXcoder(sd, cfg, tok, device)
    else:
        raise FileNotFoundError(
            f"UniXcoder checkpoint {model!r} is not a local directory and hub downloads are unavailable: point "
            "CODERAG_HIP_WEIGHTS (or ProviderConfig.model) at a local copy of microsoft/unixcoder-base, or pass "
            "extra={'synthetic_weights': <seed>} for seeded stand-in weights")
    _MODELS[key] = m
    return m
