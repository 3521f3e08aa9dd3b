"""GPU parity of the HIP encoder kernels: each kernel against a plain PyTorch fp32 computation of the same op on the
same bf16-rounded inputs, then the whole 12-layer forward against the HF-pinned fixtures.  Tolerances are those of
bf16 storage (8 mantissa bits) with f32 accumulation; they are stated at each assert."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _env():
    import torch
    import coderag_amd  # noqa: F401
    from coderag_amd import ffi
    return torch, ffi, torch.device("cuda:0")


def _close(torch, got, ref, rel, abs_):
    err = (got.float() - ref.float()).abs()
    lim = rel * ref.float().abs() + abs_
    assert bool((err <= lim).all()), f"max err {err.max().item():.4g} (worst excess {(err - lim).max().item():.4g})"


# T >= 3841 rows exercises the per-XCD super-tile order (>= 16 panels), smaller T the linear order; ragged T the row guards
# everything below a few thousand rows goes to k_gemm_mid (the cost model in crh_encoder.hip), a single row included
@pytest.mark.parametrize("T,N,K,act", [(1, 768, 768, 0), (16, 2304, 768, 0), (9, 3072, 768, 1), (16, 768, 3072, 0), (2, 768, 256, 1), (17, 2304, 768, 0), (64, 3072, 768, 1), (33, 768, 3072, 0), (48, 768, 256, 1), (65, 2304, 768, 0), (200, 3072, 768, 1), (256, 768, 3072, 0), (500, 2304, 768, 0), (513, 768, 768, 0),
                                       (384, 768, 768, 0), (200, 2304, 768, 0), (130, 3072, 768, 1), (256, 768, 3072, 0), (500, 2304, 768, 0), (513, 768, 768, 0),
                                       (5000, 2304, 768, 0), (9300, 3072, 768, 1), (4097, 768, 3072, 0)])
def test_gemm_bias_act(gpu, T, N, K, act):
    torch, ffi, dev = _env()
    g = torch.Generator(device="cpu").manual_seed(T + N)
    a = torch.randn((T, K), generator=g).to(dev, torch.bfloat16)
    w = (torch.randn((N, K), generator=g) / K ** 0.5).to(dev, torch.bfloat16)
    b = torch.randn((N,), generator=g).to(dev)
    y = torch.empty((T, N), dtype=torch.bfloat16, device=dev)
    ffi.check(ffi.lib().crh_gemm_bf16_bias(a.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), T, N, K, act, 0))
    ref = a.float() @ w.float().T + b
    if act:
        ref = torch.nn.functional.gelu(ref)             # erf form
    torch.cuda.synchronize()
    _close(torch, y, ref, rel=2 ** -7, abs_=2e-3)           # one bf16 rounding of the result + f32 accumulation order


# k_gemm_mid (64x64 tiles, 4-stage LDS-DMA ring): full and ragged last row tiles, every epilogue
@pytest.mark.parametrize("T,N,K,act", [(600, 768, 768, 0), (1000, 2304, 768, 0), (2048, 3072, 768, 1), (1537, 768, 3072, 0), (4096, 768, 3072, 0),
                                       (4095, 2304, 768, 1), (577, 128, 256, 1), (2049, 3072, 768, 0)])
def test_gemm_mid_shapes(gpu, T, N, K, act):
    torch, ffi, dev = _env()
    g = torch.Generator(device="cpu").manual_seed(T + N)
    a = torch.randn((T, K), generator=g).to(dev, torch.bfloat16)
    w = (torch.randn((N, K), generator=g) / K ** 0.5).to(dev, torch.bfloat16)
    b = torch.randn((N,), generator=g).to(dev)
    y = torch.full((T + 8, N), 7.0, dtype=torch.bfloat16, device=dev)     # 8 guard rows: nothing may be written past T
    ffi.check(ffi.lib().crh_gemm_bf16_bias(a.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), T, N, K, act, 0))
    ref = a.float() @ w.float().T + b
    if act:
        ref = torch.nn.functional.gelu(ref)
    torch.cuda.synchronize()
    _close(torch, y[:T], ref, rel=2 ** -7, abs_=2e-3)
    assert bool((y[T:] == 7.0).all())


# The 256x256 ping-pong kernel (crh_gemm256.hpp) takes over from 256 tiles up: large T through the public entry, and
# small / ragged / single-tile shapes through the debug entry (variant 16 runs it regardless of the tile count).
@pytest.mark.parametrize("T,N,K,act", [(22000, 768, 768, 0), (8192, 3072, 768, 1), (7400, 2304, 768, 0), (21931, 768, 3072, 0)])
def test_gemm256_through_public_entry(gpu, T, N, K, act):
    test_gemm_bias_act(gpu, T, N, K, act)


@pytest.mark.parametrize("T,N,K", [(256, 256, 256), (1, 256, 256), (300, 768, 768), (777, 2304, 384), (4096, 768, 3072), (5121, 3072, 768)])
def test_gemm256_small_and_ragged(gpu, T, N, K):
    torch, ffi, dev = _env()
    g = torch.Generator(device="cpu").manual_seed(T * 7 + N)
    a = torch.randn((T, K), generator=g).to(dev, torch.bfloat16)
    w = (torch.randn((N, K), generator=g) / K ** 0.5).to(dev, torch.bfloat16)
    b = torch.randn((N,), generator=g).to(dev)
    y = torch.full((T + 8, N), 7.0, dtype=torch.bfloat16, device=dev)     # 8 guard rows: nothing may be written past T
    ffi.check(ffi.debug_lib().crh_debug_gemm_variant(a.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), T, N, K, 16, 0))
    ref = a.float() @ w.float().T + b
    torch.cuda.synchronize()
    _close(torch, y[:T], ref, rel=2 ** -7, abs_=2e-3)
    assert bool((y[T:] == 7.0).all())


def test_gemm256_repeatable(gpu):
    """The schedule is ordered by counted waits and barrier ticks, not by luck: 20 runs of a many-tile shape are
    bit-identical to the first (a read that raced its LDS-DMA would show up as an occasional differing tile)."""
    torch, ffi, dev = _env()
    T, N, K = 24576, 768, 3072
    g = torch.Generator(device="cpu").manual_seed(5)
    a = torch.randn((T, K), generator=g).to(dev, torch.bfloat16)
    w = (torch.randn((N, K), generator=g) / K ** 0.5).to(dev, torch.bfloat16)
    b = torch.randn((N,), generator=g).to(dev)
    outs = []
    for _ in range(20):
        y = torch.empty((T, N), dtype=torch.bfloat16, device=dev)
        ffi.check(ffi.lib().crh_gemm_bf16_bias(a.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), T, N, K, 0, 0))
        outs.append(y)
    torch.cuda.synchronize()
    ref = a.float() @ w.float().T + b
    _close(torch, outs[0], ref, rel=2 ** -7, abs_=2e-3)
    for y in outs[1:]:
        assert torch.equal(y.view(torch.int16), outs[0].view(torch.int16))


@pytest.mark.parametrize("T,N,K,act", [(3000, 768, 3072, 0), (2049, 3072, 768, 1), (64, 768, 3072, 0)])
def test_gemm_mid_repeatable_and_equal_to_the_other_tiled_kernels(gpu, T, N, K, act):
    """k_gemm_mid's 4-stage ring is ordered by counted waits and one barrier per step: 30 runs are bit-identical.  And the
    three tiled kernels add the same products in the same order (one 16x16x32 MFMA chain along K per output block), so
    k_gemm_nt and the ping-pong kernel (debug variants 0 and 16) give the very same bits where the epilogue is the same (bias only)."""
    torch, ffi, dev = _env()
    g = torch.Generator(device="cpu").manual_seed(T)
    a = torch.randn((T, K), generator=g).to(dev, torch.bfloat16)
    w = (torch.randn((N, K), generator=g) / K ** 0.5).to(dev, torch.bfloat16)
    b = torch.randn((N,), generator=g).to(dev)
    outs = []
    for _ in range(30):
        y = torch.empty((T, N), dtype=torch.bfloat16, device=dev)
        ffi.check(ffi.lib().crh_gemm_bf16_bias(a.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), T, N, K, act, 0))
        outs.append(y)
    torch.cuda.synchronize()
    for y in outs[1:]:
        assert torch.equal(y.view(torch.int16), outs[0].view(torch.int16))
    if not act:
        y = torch.empty((T, N), dtype=torch.bfloat16, device=dev)
        for variant in (0, 16):                               # 0: k_gemm_nt, 16: the 256x256 ping-pong kernel
            y.fill_(0)
            ffi.check(ffi.debug_lib().crh_debug_gemm_variant(a.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), T, N, K, variant, 0))
            torch.cuda.synchronize()
            assert torch.equal(y.view(torch.int16), outs[0].view(torch.int16)), variant


@pytest.mark.parametrize("T,K", [(1, 768), (40, 3072), (64, 768), (129, 3072), (256, 768), (257, 768), (300, 3072), (6000, 768), (22100, 3072), (23040, 768),
                                 (513, 768), (1000, 3072), (2048, 3072), (4095, 768), (4096, 3072), (16, 3072), (17, 768)])
def test_gemm_residual_layernorm(gpu, T, K):
    torch, ffi, dev = _env()
    g = torch.Generator(device="cpu").manual_seed(K)
    a = torch.randn((T, K), generator=g).to(dev, torch.bfloat16)
    w = (torch.randn((768, K), generator=g) / K ** 0.5).to(dev, torch.bfloat16)
    b = torch.randn((768,), generator=g).to(dev)
    res = torch.randn((T, 768), generator=g).to(dev, torch.bfloat16)
    gam = (1 + 0.1 * torch.randn((768,), generator=g)).to(dev)
    bet = (0.1 * torch.randn((768,), generator=g)).to(dev)
    y = torch.empty((T, 768), dtype=torch.bfloat16, device=dev)
    ffi.check(ffi.lib().crh_gemm_bf16_bias_res_ln(a.data_ptr(), w.data_ptr(), b.data_ptr(), res.data_ptr(), gam.data_ptr(),
                                                  bet.data_ptr(), 1e-5, y.data_ptr(), T, 768, K, 0))
    pre = a.float() @ w.float().T + b + res.float()
    ref = torch.nn.functional.layer_norm(pre, (768,), gam, bet, 1e-5)
    torch.cuda.synchronize()
    _close(torch, y, ref, rel=2 ** -6, abs_=2e-2)           # the pre-LN sum is held in bf16 before normalisation


# ---- LayerNorm folded into the GEMMs around it (round 5; csrc/crh_encoder.hip "LayerNorm folded ...", modeling_roberta.py:329-340,387-398)
# T picks the kernel: a few rows -> k_gemm_mid, thousands -> k_gemm_nt, tens of thousands -> the 256x256 ping-pong kernel; ragged T the guards
@pytest.mark.parametrize("with_stats", [True, False])
@pytest.mark.parametrize("T,K", [(1, 768), (40, 3072), (257, 768), (1000, 3072), (4095, 768), (6000, 768), (22100, 3072), (23040, 768), (66000, 768)])
def test_folded_producer_gemm_residual_and_statistics(gpu, T, K, with_stats):
    """crh_gemm_bf16_res_lnstats: y = bf16(x @ w^T + bias + h), h = (residual * rstd + nmr) * gamma (or the residual as it is), and
    the (rstd, nmr) of the rows of y AS STORED."""
    torch, ffi, dev = _env()
    g = torch.Generator(device="cpu").manual_seed(K + T)
    a = torch.randn((T, K), generator=g).to(dev, torch.bfloat16)
    w = (torch.randn((768, K), generator=g) / K ** 0.5).to(dev, torch.bfloat16)
    b = torch.randn((768,), generator=g).to(dev)
    res = (3.0 * torch.randn((T, 768), generator=g) + 0.7).to(dev, torch.bfloat16)       # un-normalised rows: some scale, some mean
    gam = (1 + 0.1 * torch.randn((768,), generator=g)).to(dev)
    mu = res.float().mean(-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(res.float().var(-1, unbiased=False, keepdim=True) + 1e-5)
    rst = torch.cat([rstd, -mu * rstd], 1).contiguous()
    y = torch.full((T + 4, 768), 7.0, dtype=torch.bfloat16, device=dev)
    part = torch.empty((T, 24, 2), dtype=torch.float32, device=dev)
    st = torch.full((T + 4, 2), 7.0, dtype=torch.float32, device=dev)
    ffi.check(ffi.lib().crh_gemm_bf16_res_lnstats(a.data_ptr(), w.data_ptr(), b.data_ptr(), res.data_ptr(), rst.data_ptr() if with_stats else None,
                                                  gam.data_ptr() if with_stats else None, 1e-5, y.data_ptr(), part.data_ptr(), st.data_ptr(), T, 768, K, 0))
    h = (res.float() * rst[:, :1] + rst[:, 1:]) * gam if with_stats else res.float()
    ref = a.float() @ w.float().T + b + h
    torch.cuda.synchronize()
    _close(torch, y[:T], ref, rel=2 ** -7, abs_=4e-3)
    assert bool((y[T:] == 7.0).all()) and bool((st[T:] == 7.0).all())                   # nothing past T
    yy = y[:T].float()                                                                   # the statistics describe what was stored
    mu_y = yy.mean(-1, keepdim=True)
    rstd_y = 1.0 / torch.sqrt(yy.var(-1, unbiased=False, keepdim=True) + 1e-5)
    assert torch.allclose(st[:T, :1], rstd_y, rtol=2e-5, atol=0) and torch.allclose(st[:T, 1:], -mu_y * rstd_y, rtol=2e-5, atol=2e-6)
    if not with_stats:      # without statistics this is epilogue 2 (bias + residual, one rounding): the same bits as the old entry point's pre-LN sum
        y2 = torch.empty((T, 768), dtype=torch.bfloat16, device=dev)
        one, zero = torch.ones((768,), device=dev), torch.zeros((768,), device=dev)
        if K > 1024:        # (that entry point adds the residual in the GEMM epilogue only for K > 1024; its LayerNorm then runs in place)
            ffi.check(ffi.lib().crh_gemm_bf16_bias_res_ln(a.data_ptr(), w.data_ptr(), b.data_ptr(), res.data_ptr(), one.data_ptr(), zero.data_ptr(), 1e-5,
                                                          y2.data_ptr(), T, 768, K, 0))
            torch.cuda.synchronize()
            _close(torch, y2, torch.nn.functional.layer_norm(yy, (768,), one, zero, 1e-5), rel=2 ** -7, abs_=4e-3)


@pytest.mark.parametrize("T,N,act", [(1, 2304, 0), (33, 3072, 1), (500, 2304, 0), (2048, 3072, 1), (5000, 2304, 0), (9300, 3072, 1), (22000, 2304, 0), (66000, 3072, 1)])
def test_folded_consumer_gemm_finishes_the_layernorm(gpu, T, N, act):
    """crh_gemm_bf16_lnin on un-normalised rows == the plain GEMM on their LayerNorm, up to bf16 rounding of the gain-scaled weights."""
    torch, ffi, dev = _env()
    K = 768
    g = torch.Generator(device="cpu").manual_seed(N + T)
    r = (2.5 * torch.randn((T, K), generator=g) - 0.4).to(dev, torch.bfloat16)
    w = (torch.randn((N, K), generator=g) / K ** 0.5).to(dev)
    b = torch.randn((N,), generator=g).to(dev)
    gam = (1 + 0.1 * torch.randn((K,), generator=g)).to(dev)
    bet = (0.1 * torch.randn((K,), generator=g)).to(dev)
    ws = (w * gam[None, :]).to(torch.bfloat16)
    colsum = ws.double().sum(1).float().contiguous()
    bias_f = (b.double() + w.double() @ bet.double()).float().contiguous()
    mu = r.float().mean(-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(r.float().var(-1, unbiased=False, keepdim=True) + 1e-5)
    rst = torch.cat([rstd, -mu * rstd], 1).contiguous()
    y = torch.full((T + 4, N), 7.0, dtype=torch.bfloat16, device=dev)
    ffi.check(ffi.lib().crh_gemm_bf16_lnin(r.data_ptr(), rst.data_ptr(), ws.data_ptr(), colsum.data_ptr(), bias_f.data_ptr(), y.data_ptr(), T, N, K, act, 0))
    ref = torch.nn.functional.layer_norm(r.float(), (K,), gam, bet, 1e-5) @ w.T + b
    if act:
        ref = torch.nn.functional.gelu(ref)
    torch.cuda.synchronize()
    _close(torch, y[:T], ref, rel=2 ** -6, abs_=1.5e-2)       # + the gain-scaled weights' own bf16 rounding over K = 768 products
    assert bool((y[T:] == 7.0).all())


def test_a_rows_folded_results_do_not_depend_on_the_batch_it_sits_in(gpu):
    """The three tiled kernels build a row's statistics from the same per-lane sums joined the same way: a row's output, its statistics
    and what the consumer makes of them are bit-identical whether the row is computed among 40 rows (k_gemm_mid), 3 000 (k_gemm_nt)
    or 30 000 (the ping-pong kernel)."""
    torch, ffi, dev = _env()
    g = torch.Generator(device="cpu").manual_seed(5)
    Tbig, K = 30000, 768
    a = torch.randn((Tbig, K), generator=g).to(dev, torch.bfloat16)
    w = (torch.randn((768, K), generator=g) / K ** 0.5).to(dev, torch.bfloat16)
    b = torch.randn((768,), generator=g).to(dev)
    res = (2.0 * torch.randn((Tbig, 768), generator=g) + 0.3).to(dev, torch.bfloat16)
    gam = (1 + 0.1 * torch.randn((768,), generator=g)).to(dev)
    mu = res.float().mean(-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(res.float().var(-1, unbiased=False, keepdim=True) + 1e-5)
    rst = torch.cat([rstd, -mu * rstd], 1).contiguous()
    w2 = (torch.randn((2304, 768), generator=g) / 768 ** 0.5).to(dev, torch.bfloat16)
    c2, b2 = w2.double().sum(1).float().contiguous(), torch.randn((2304,), generator=g).to(dev)
    outs = []
    for T in (40, 3000, Tbig):
        y = torch.empty((T, 768), dtype=torch.bfloat16, device=dev)
        part = torch.empty((T, 24, 2), dtype=torch.float32, device=dev)
        st = torch.empty((T, 2), dtype=torch.float32, device=dev)
        ffi.check(ffi.lib().crh_gemm_bf16_res_lnstats(a.data_ptr(), w.data_ptr(), b.data_ptr(), res.data_ptr(), rst.data_ptr(), gam.data_ptr(), 1e-5,
                                                      y.data_ptr(), part.data_ptr(), st.data_ptr(), T, 768, K, 0))
        z = torch.empty((T, 2304), dtype=torch.bfloat16, device=dev)
        ffi.check(ffi.lib().crh_gemm_bf16_lnin(y.data_ptr(), st.data_ptr(), w2.data_ptr(), c2.data_ptr(), b2.data_ptr(), z.data_ptr(), T, 2304, 768, 0, 0))
        torch.cuda.synchronize()
        outs.append((y[:40].clone(), st[:40].clone(), z[:40].clone()))
    for y, st, z in outs[1:]:
        assert torch.equal(y.view(torch.int16), outs[0][0].view(torch.int16))
        assert torch.equal(st.view(torch.int32), outs[0][1].view(torch.int32))
        assert torch.equal(z.view(torch.int16), outs[0][2].view(torch.int16))


def _kmask(torch, valid):                                   # valid: bool [B, L] -> int64 [B, ceil(L/64)] bit words
    B, L = valid.shape
    Lp = (L + 63) // 64 * 64
    v = torch.zeros((B, Lp), dtype=torch.bool)
    v[:, :L] = valid
    w = (v.reshape(B, Lp // 64, 64).to(torch.int64) << torch.arange(64, dtype=torch.int64)).sum(-1)
    return w.contiguous()


@pytest.mark.parametrize("B,L", [(3, 64), (2, 192), (4, 512), (5, 16), (3, 80), (2, 208), (2, 496)])
def test_attention(gpu, B, L):
    torch, ffi, dev = _env()
    H = 12
    g = torch.Generator(device="cpu").manual_seed(B * L)
    qkv = torch.randn((B, L, 3 * H * 64), generator=g).to(dev, torch.bfloat16)
    lens = torch.randint(5, L + 1, (B,), generator=g)
    lens[0] = L
    valid = torch.arange(L)[None, :] < lens[:, None]
    if L > 64:
        valid[1, 70:75] = False                              # interior masked keys (ids == pad inside the text)
    km = _kmask(torch, valid).to(dev)
    out = torch.full((B, L, H * 64), float("nan"), dtype=torch.bfloat16, device=dev)
    ffi.check(ffi.lib().crh_attn_fwd_varlen(qkv.data_ptr(), km.data_ptr(), out.data_ptr(), B, L, H, 0))
    q, k, v = (t.reshape(B, L, H, 64).transpose(1, 2).float() for t in qkv.split(H * 64, dim=-1))
    s = q @ k.transpose(-1, -2) * 0.125
    s = s.masked_fill(~valid.to(dev)[:, None, None, :], float("-inf"))
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, L, H * 64)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out.float()).all()), "every output row must be finite (pad rows included)"
    last = [int(torch.nonzero(valid[b])[-1]) + 1 for b in range(B)]
    for b in range(B):                                       # rows up to the last real token are specified
        _close(torch, out[b, : last[b]], ref[b, : last[b]], rel=2 ** -6, abs_=1.5e-2)   # P and V are bf16 in the PV product


def test_embed_ln_and_pool(gpu):
    torch, ffi, dev = _env()
    B, L, D, V, pad = 5, 144, 768, 2000, 1
    g = torch.Generator(device="cpu").manual_seed(9)
    word = torch.randn((V, D), generator=g).to(dev, torch.bfloat16)
    pos = torch.randn((L + 2, D), generator=g).to(dev, torch.bfloat16)
    typ = torch.randn((D,), generator=g).to(dev, torch.bfloat16)
    gam = (1 + 0.1 * torch.randn((D,), generator=g)).to(dev)
    bet = (0.1 * torch.randn((D,), generator=g)).to(dev)
    ids = torch.randint(3, V, (B, L), generator=g, dtype=torch.int32)
    for b, n in enumerate((144, 7, 64, 65, 100)):
        ids[b, n:] = pad
    ids[2, 10] = pad                                         # an interior pad: position ids must skip it
    ids_d = ids.to(dev)
    out = torch.empty((B, L, D), dtype=torch.bfloat16, device=dev)
    km = torch.empty((B, (L + 63) // 64), dtype=torch.int64, device=dev)
    ffi.check(ffi.lib().crh_embed_ln(ids_d.data_ptr(), word.data_ptr(), pos.data_ptr(), typ.data_ptr(), gam.data_ptr(), bet.data_ptr(),
                                     1e-5, pad, out.data_ptr(), km.data_ptr(), B, L, D, 0))
    valid = ids_d != pad
    pid = torch.cumsum(valid.long(), 1) * valid.long() + pad
    ref = torch.nn.functional.layer_norm((word[ids_d.long()].float() + typ.float()) + pos[pid].float(), (D,), gam, bet, 1e-5)
    torch.cuda.synchronize()
    assert torch.equal(km.cpu(), _kmask(torch, valid.cpu()))
    _close(torch, out, ref, rel=2 ** -7, abs_=4e-3)
    sent = torch.empty((B, D), dtype=torch.float32, device=dev)
    ffi.check(ffi.lib().crh_masked_mean_pool(out.data_ptr(), km.data_ptr(), sent.data_ptr(), B, L, D, 0))
    m = valid.float()
    ref_s = (out.float() * m[..., None]).sum(1) / m.sum(-1, keepdim=True)
    torch.cuda.synchronize()
    _close(torch, sent, ref_s, rel=1e-5, abs_=1e-5)          # f32 sums of the same bf16 values, order differs


# per fixture: (min cosine, max relative L2) vs the bf16-storage oracle, and vs the fp32 HF fixture
ENCODER_TOL = {"tiny": ((0.9999, 1e-2), (0.9995, 3e-2)),
               "base": ((0.9993, 4e-2), (0.997, 8e-2)),
               "hfinit": ((0.99995, 1e-2), (0.9999, 1.5e-2)),
               "hfln": ((0.99995, 1e-2), (0.9999, 1.5e-2))}      # HF-init matrices, the sharp fixture's biases and LayerNorm parameters (round 5)


@pytest.mark.parametrize("form", ["two LayerNorm kernels", "ln_fold", "residual_f32"])
@pytest.mark.parametrize("name", ["tiny", "base", "hfinit", "hfln"])
def test_full_encoder_against_hf_fixture(gpu, name, form):
    """Whole forward against (a) the oracle in its ``bf16_storage`` mode -- f32 arithmetic, bf16 rounding exactly where the
    kernels store bf16 -- and (b) the fp32 HF fixture; tolerances per fixture in ENCODER_TOL.
    ``base`` carries deliberately SHARP weights (O(1) activations, 2/sqrt(H) Q/K scale: attention far from uniform, every
    rounding amplified through 12 layers): bf16 WEIGHTS ALONE put the f32-arithmetic oracle 3.2e-2 / cosine 0.9995 from HF
    there, and bf16 activations 5.2-5.9e-2 (tests/test_precision_budget.py pins that budget on CPU), so (b) cannot go below
    that without f32 weights.  ``hfinit`` is the same geometry with HF-init statistics (N(0, 0.02^2) matrices, unit
    LayerNorm): the whole bf16 pipeline stays within ~5e-3 of HF there; real checkpoints sit between the two."""
    torch, ffi, dev = _env()
    from coderag_amd import encoder as drv
    from oracle import encoder as orc
    z = np.load(os.path.join(GOLD, f"encoder_{name}.npz"))
    init = str(z["init"]) if "init" in z.files else "sharp"
    c = [int(v) for v in z["cfg"]]
    kw = dict(vocab_size=c[0], hidden_size=c[1], num_layers=c[2], num_heads=c[3], intermediate_size=c[4],
              max_position_embeddings=c[5], type_vocab_size=c[6], pad_token_id=c[7], layer_norm_eps=float(z["eps"]))
    fold, res32 = form == "ln_fold", form == "residual_f32"     # the default and the two opt-in forms of the LayerNorm step (EncoderConfig)
    cfg = drv.EncoderConfig(**kw, ln_fold=fold, residual_f32=res32)
    weights = drv.synthetic_weights(cfg, int(z["seed"]), init=init)
    model = drv.HipUniXcoder(weights, cfg, drv.HashTokenizer(cfg.vocab_size), 0)
    ids = torch.from_numpy(z["ids"].astype(np.int32)).to(dev)
    got = model.forward_ids(ids).cpu().numpy()

    def dist(ref):
        cos = (got * ref).sum(1) / (np.linalg.norm(got, axis=1) * np.linalg.norm(ref, axis=1))
        return cos.min(), (np.linalg.norm(got - ref, axis=1) / np.linalg.norm(ref, axis=1)).max()
    cos_a, rel_a = dist(orc.forward(weights, orc.EncoderConfig(**kw), z["ids"], bf16_storage=True, ln_fold=fold, residual_f32=res32))
    cos_b, rel_b = dist(z["sent"])
    print(f"encoder[{name}] vs bf16-storage oracle: cos {cos_a:.6f} rel {rel_a:.4f}; vs HF fp32: cos {cos_b:.6f} rel {rel_b:.4f}")
    if os.environ.get("CODERAG_TEST_REPORT"):      # (a report file only on request: CODERAG_TEST_REPORT=<path>)
        with open(os.environ["CODERAG_TEST_REPORT"], "a") as f:
            f.write(f"{name}: vs bf16-storage oracle cos {cos_a:.6f} rel {rel_a:.5f}; vs HF fp32 cos {cos_b:.6f} rel {rel_b:.5f}\n")
    (ca, ra), (cb, rb_) = ENCODER_TOL[name]
    assert cos_a >= ca and rel_a <= ra, (cos_a, rel_a)
    assert cos_b >= cb and rel_b <= rb_, (cos_b, rel_b)
    # quirk Q1: ragged id lists through the length-bucketing driver give the same vectors as the padded batch
    lists = [row[: int((row != cfg.pad_token_id).sum())].tolist() for row in z["ids"] if cfg.pad_token_id not in row[: int((row != cfg.pad_token_id).sum())]]
    keep = [i for i, row in enumerate(z["ids"]) if cfg.pad_token_id not in row[: int((row != cfg.pad_token_id).sum())]]
    again = model.embed_ids(lists).cpu().numpy()
    assert np.array_equal(again, got[keep])


@pytest.mark.parametrize("form", ["two LayerNorm kernels", "ln_fold", "residual_f32"])
@pytest.mark.parametrize("name", ["tiny", "base", "hfinit", "hfln"])
def test_packed_forward_against_hf_fixture(gpu, name, form):
    """The PACKED forward -- what embed_ids / embed_texts / the provider / bench.py run -- fed the fixtures' rows directly
    (tokens back to back, row offsets) against the bf16-storage oracle and the fp32 HF vectors, at ENCODER_TOL: pinned by
    the fixtures themselves, not through the padded forward.  Rows with an interior pad token are part of it."""
    torch, ffi, dev = _env()
    from coderag_amd import encoder as drv
    from oracle import encoder as orc
    z = np.load(os.path.join(GOLD, f"encoder_{name}.npz"))
    init = str(z["init"]) if "init" in z.files else "sharp"
    c = [int(v) for v in z["cfg"]]
    kw = dict(vocab_size=c[0], hidden_size=c[1], num_layers=c[2], num_heads=c[3], intermediate_size=c[4],
              max_position_embeddings=c[5], type_vocab_size=c[6], pad_token_id=c[7], layer_norm_eps=float(z["eps"]))
    fold, res32 = form == "ln_fold", form == "residual_f32"     # the default and the two opt-in forms of the LayerNorm step (EncoderConfig)
    cfg = drv.EncoderConfig(**kw, ln_fold=fold, residual_f32=res32)
    weights = drv.synthetic_weights(cfg, int(z["seed"]), init=init)
    model = drv.HipUniXcoder(weights, cfg, drv.HashTokenizer(cfg.vocab_size), 0)
    rows = []
    for row in z["ids"]:
        real = np.flatnonzero(row != cfg.pad_token_id)
        rows.append(row[: int(real[-1]) + 1].astype(np.int32))       # up to the last real token (interior pads stay)
    flat, off, Lmax = model.pack_rows(rows, list(range(len(rows))))
    got = model.forward_packed(torch.from_numpy(flat).to(dev), torch.from_numpy(off).to(dev), Lmax, verify=True).cpu().numpy()

    def dist(ref):
        cos = (got * ref).sum(1) / (np.linalg.norm(got, axis=1) * np.linalg.norm(ref, axis=1))
        return cos.min(), (np.linalg.norm(got - ref, axis=1) / np.linalg.norm(ref, axis=1)).max()
    cos_a, rel_a = dist(orc.forward(weights, orc.EncoderConfig(**kw), z["ids"], bf16_storage=True, ln_fold=fold, residual_f32=res32))
    cos_b, rel_b = dist(z["sent"])
    print(f"packed encoder[{name}] vs bf16-storage oracle: cos {cos_a:.6f} rel {rel_a:.4f}; vs HF fp32: cos {cos_b:.6f} rel {rel_b:.4f}")
    (ca, ra), (cb, rb_) = ENCODER_TOL[name]
    assert cos_a >= ca and rel_a <= ra, (cos_a, rel_a)
    assert cos_b >= cb and rel_b <= rb_, (cos_b, rel_b)


def test_packed_entry_points_reject_bad_offsets_without_faulting(gpu):
    """row_off is device data the library cannot inspect at call time: the kernels clamp every row to the T tokens of the
    buffers, and the device-side check reports the array as CRH_E_INVALID -- an error code, not a GPU fault -- at
    crh_encoder_finish (or at the next packed call once the check has run).  Guard rows past T stay untouched."""
    torch, ffi, dev = _env()
    from coderag_amd import encoder as drv
    cfg = drv.EncoderConfig(num_layers=1)
    model = drv.HipUniXcoder(drv.synthetic_weights(cfg, 3), cfg, drv.HashTokenizer(cfg.vocab_size), 0)
    L_, H = ffi.lib(), 12
    T, B, Lmax = 300, 4, 128
    rng = np.random.default_rng(0)
    ids = torch.from_numpy(rng.integers(3, cfg.vocab_size, T).astype(np.int32)).to(dev)
    bad_sets = {"beyond T": [0, 100, 5000, 100000, 300], "decreasing": [0, 200, 100, 250, 300], "negative": [0, -50, 100, 200, 300],
                "row longer than Lmax": [0, 10, 20, 290, 300], "does not start at 0": [7, 100, 200, 250, 300], "does not end at T": [0, 100, 200, 250, 280],
                "garbage": [-2**31, 2**31 - 1, -1, 2**30, 17]}
    for what, offs in bad_sets.items():
        off = torch.tensor(offs, dtype=torch.int32, device=dev)
        x = torch.full((T + 64, 768), 7.0, dtype=torch.bfloat16, device=dev)             # 64 guard rows behind every buffer
        qkv = torch.randn((T + 64, 3 * 768), device=dev).to(torch.bfloat16)
        ctx = torch.full((T + 64, 768), 7.0, dtype=torch.bfloat16, device=dev)
        km = torch.zeros((B, (Lmax + 63) // 64), dtype=torch.int64, device=dev)
        sent = torch.zeros((B, 768), dtype=torch.float32, device=dev)
        ffi.check(L_.crh_embed_ln_packed(ids.data_ptr(), off.data_ptr(), *model._emb_ptrs, 1e-5, cfg.pad_token_id, x.data_ptr(), km.data_ptr(), B, T, Lmax, 768, 0))
        rc2 = L_.crh_attn_fwd_packed(qkv.data_ptr(), off.data_ptr(), km.data_ptr(), ctx.data_ptr(), B, T, Lmax, H, 0)
        rc3 = L_.crh_masked_mean_pool_packed(x.data_ptr(), off.data_ptr(), km.data_ptr(), sent.data_ptr(), B, T, Lmax, 768, 0)
        rc4 = L_.crh_encoder_finish(0)
        torch.cuda.synchronize()                                                          # no fault: the device is alive
        codes = [rc2, rc3, rc4]
        assert codes.count(ffi.E_INVALID) == 1 and all(c in (ffi.OK, ffi.E_INVALID) for c in codes), (what, codes)
        assert b"row_off" in L_.crh_last_error()
        assert bool((x[T:] == 7.0).all()) and bool((ctx[T:] == 7.0).all()), what
        assert L_.crh_encoder_finish(0) == ffi.OK                                         # one report per offence
    # a good array after the bad ones: accepted, and the forward still equals the padded one
    with pytest.raises(ffi.NativeError):
        model.forward_packed(ids, torch.tensor([0, 100, 90, 300], dtype=torch.int32, device=dev), 128, verify=True)
    good = torch.tensor([0, 100, 228, 300], dtype=torch.int32, device=dev)
    got = model.forward_packed(ids, good, 128, verify=True).cpu().numpy()
    pad = np.full((3, 128), cfg.pad_token_id, np.int32)
    h = ids.cpu().numpy()
    for r, (a, b) in enumerate(((0, 100), (100, 228), (228, 300))):
        pad[r, : b - a] = h[a:b]
    assert np.array_equal(got, model.forward_ids(torch.from_numpy(pad).to(dev)).cpu().numpy())


def test_provider_end_to_end(gpu):
    """HipUniXcoderProvider through the reference's provider surface: embed / embed_batch on ragged texts."""
    import asyncio
    _env()
    from coderag_amd.providers import HipUniXcoderProvider, ProviderConfig
    p = HipUniXcoderProvider(ProviderConfig(provider="unixcoder-hip", model="synthetic", extra={"synthetic_weights": 3, "num_layers": 2}))
    texts = ["def add(a, b):\n    return a + b", "class Foo:\n    pass", "x = 1", "def add(a, b):\n    return a + b"]

    async def go():
        one = await p.embed(texts[0])
        many = await p.embed_batch(texts, batch_size=3)
        return one, many
    one, many = asyncio.run(go())
    assert p.embedding_dim == 768 and len(one) == 768 and isinstance(one[0], float) and len(many) == 4
    # same text in different batches: the same vector, to the bit (the reference's forward is batch-invariant; so is this one)
    assert many[0] == many[3] and one == many[0]
    assert not np.allclose(many[0], many[1], atol=1e-3)


@pytest.mark.parametrize("form", ["residual_f32", "ln_fold"])
def test_provider_extras_select_the_opt_in_forms_of_the_layernorm_step(gpu, form):
    """``ProviderConfig.extra={"residual_f32": True}`` / ``{"ln_fold": True}``: the fidelity levers per provider,
    separate singletons from the default model, embeddings within bf16 distance of the default's."""
    import asyncio
    _env()
    from coderag_amd.providers import HipUniXcoderProvider, ProviderConfig
    base = HipUniXcoderProvider(ProviderConfig(provider="unixcoder-hip", model="synthetic", extra={"synthetic_weights": 3, "num_layers": 2}))
    lever = HipUniXcoderProvider(ProviderConfig(provider="unixcoder-hip", model="synthetic", extra={"synthetic_weights": 3, "num_layers": 2, form: True}))
    texts = ["def add(a, b):\n    return a + b", "class Foo:\n    pass", "x = 1"]
    a = np.asarray(asyncio.run(base.embed_batch(texts)))
    b = np.asarray(asyncio.run(lever.embed_batch(texts)))
    assert getattr(lever._load().cfg, form) is True and getattr(base._load().cfg, form) is False
    assert lever._load() is not base._load()
    cos = (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
    assert cos.min() > 0.999 and not np.array_equal(a, b)


def test_checkpoint_directory_with_native_tokenizer(gpu, tmp_path):
    """A local checkpoint directory (config.json + model.safetensors + vocab.json + merges.txt, as microsoft/unixcoder-base
    ships them; built here from seeded weights and a locally trained vocabulary) loads through load_unixcoder with the
    native byte-level BPE, and embed_texts() equals: HF tokenizer ids -> the reference's wrapping -> embed_ids()."""
    import glob
    import json
    torch, ffi, dev = _env()
    from safetensors.torch import save_file
    from tokenizers import ByteLevelBPETokenizer
    from transformers import RobertaTokenizer
    from coderag_amd import encoder as drv
    from coderag_amd.tokenizer_native import NativeBpeTokenizer

    d = str(tmp_path)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "code-rag_amd", "*.py")))
    tr = ByteLevelBPETokenizer(add_prefix_space=False)
    tr.train(files, vocab_size=3000, min_frequency=2,
             special_tokens=["<s>", "<pad>", "</s>", "<unk>", "<mask>", "<encoder-only>", "<decoder-only>", "<encoder-decoder>"])
    tr.save_model(d)
    cfg = drv.EncoderConfig(vocab_size=3000, num_layers=2)
    json.dump({"vocab_size": 3000, "hidden_size": 768, "num_hidden_layers": 2, "num_attention_heads": 12, "intermediate_size": 3072,
               "max_position_embeddings": 1026, "type_vocab_size": 10, "layer_norm_eps": 1e-5, "pad_token_id": 1}, open(os.path.join(d, "config.json"), "w"))
    save_file({"roberta." + k: torch.from_numpy(v) for k, v in drv.synthetic_weights(cfg, 31).items()}, os.path.join(d, "model.safetensors"))

    model = drv.load_unixcoder(d, device=0)
    assert isinstance(model.tok, NativeBpeTokenizer) and model.cfg.num_layers == 2
    texts = [open(f, encoding="utf-8").read()[i:i + 700] for f in files[:6] for i in (0, 900, 5000)] + ["", "x", "def f():\n\treturn 'ünï'  # ✓"]
    got = np.asarray(model.embed_texts(texts), dtype=np.float32)
    hf = RobertaTokenizer(os.path.join(d, "vocab.json"), os.path.join(d, "merges.txt"))
    eid = hf.convert_tokens_to_ids("<encoder-only>")
    ids = [[hf.cls_token_id, eid, hf.sep_token_id] + hf.convert_tokens_to_ids(hf.tokenize(t))[:508] + [hf.sep_token_id] for t in texts]
    want = model.embed_ids(ids).cpu().numpy()
    assert got.shape == (len(texts), 768) and np.array_equal(got, want)


def test_encoder_properties_on_the_bench_mix(gpu):
    """Size-independent properties on the bench's workload shape (BASELINE configs[1]: lognormal chunk lengths, 12 layers), where
    the oracle would take minutes: (1) the same call twice is bit-identical; (2) a chunk's embedding does not depend on what
    else is in the call -- other neighbours, another padded length, another GEMM kernel (the batch's token count picks it) --
    beyond bf16 accumulation-order noise; (3) trailing pad tokens change nothing beyond the same noise; (4) rows come back
    in input order whatever the length-bucketing did."""
    torch, ffi, dev = _env()
    from coderag_amd import encoder as drv
    cfg = drv.EncoderConfig()
    model = drv.HipUniXcoder(drv.synthetic_weights(cfg, 23), cfg, drv.HashTokenizer(cfg.vocab_size), 0)
    rng = np.random.default_rng(1234)
    n = 3000
    lens = np.clip(np.round(np.exp(rng.normal(np.log(160), 0.8, n))), 8, 512).astype(int)
    ids = [[0, 6, 2] + rng.integers(3, cfg.vocab_size, int(L) - 4).tolist() + [2] for L in lens]
    ids = [[t if t != cfg.pad_token_id else 7 for t in row] for row in ids]
    full = model.embed_ids(ids).cpu().numpy()
    again = model.embed_ids(ids).cpu().numpy()
    assert np.array_equal(full, again)                                           # (1)
    assert np.isfinite(full).all() and full.shape == (n, 768)
    scale = np.abs(full).max(axis=1, keepdims=True)

    pick = rng.choice(n, 40, replace=False)
    alone = model.embed_ids([ids[i] for i in pick]).cpu().numpy()               # a small call: other kernels, other neighbours
    # (2) BIT-identical: 40 rows of 8..512 tokens packed into one ~8k-token batch run the mid-size GEMM kernels, the 65k-token
    # batches of `full` the ping-pong kernel -- the tiled kernels add the same products in the same order, the residual joins
    # at a place fixed by the operation (crh_gemm_bf16_bias_res_ln), attention / LayerNorm / pool work row by row
    cos = (alone * full[pick]).sum(1) / (np.linalg.norm(alone, axis=1) * np.linalg.norm(full[pick], axis=1))
    print(f"batch-composition: max abs diff / max |e| = {np.abs(alone - full[pick]).max() / scale[pick].max():.2e}, min cosine {cos.min():.8f}")
    assert np.array_equal(alone, full[pick])
    for i in pick[:6]:                                                           # a call of its own (the query path)
        one = model.embed_ids([ids[int(i)]]).cpu().numpy()[0]
        assert np.array_equal(one, full[i]), (len(ids[int(i)]), np.abs(one - full[i]).max())
    tiny = [[0, 6, 2] + rng.integers(3, cfg.vocab_size, m).tolist() + [2] for m in (1, 4, 9, 12)]    # the shortest rows there are
    tiny = [[t if t != cfg.pad_token_id else 7 for t in row] for row in tiny]
    together = model.embed_ids(tiny + [ids[int(pick[0])]]).cpu().numpy()
    for r, row in enumerate(tiny):
        assert np.array_equal(model.embed_ids([row]).cpu().numpy()[0], together[r]), len(row)

    short = [ids[i] for i in pick if len(ids[i]) <= 200][:8]                     # (3) explicit padding to a longer bucket
    t = np.full((len(short), 256), cfg.pad_token_id, dtype=np.int32)
    for r, row in enumerate(short):
        t[r, :len(row)] = row
    padded = model.forward_ids(torch.from_numpy(t).to(dev)).cpu().numpy()
    tight = model.embed_ids(short).cpu().numpy()
    print(f"padding: {np.abs(padded - tight).max() / np.abs(tight).max():.2e}")
    assert np.array_equal(padded, tight)                                         # pad keys weigh exactly 0, pad tokens are not pooled

    perm = rng.permutation(200)                                                  # (4)
    sub = [ids[i] for i in perm]
    got = model.embed_ids(sub).cpu().numpy()
    assert np.array_equal(got, full[perm])
    rev = model.embed_ids(sub[::-1]).cpu().numpy()
    assert np.array_equal(rev[::-1], got)                                        # same batches, other input order: bit-identical


# ---- packed rows (crh_*_packed): the forward without padding tokens
def test_packed_forward_equals_the_padded_forward(gpu):
    """Rows back to back on one token axis (attention / embedding gather / pool take row offsets, GEMMs and LayerNorms see only
    real tokens) against the padded batch: the arithmetic per token is the same, so the sentence vectors agree to accumulation
    noise (other GEMM tiles see other neighbours; the kernels themselves are the same)."""
    torch, ffi, dev = _env()
    from coderag_amd import encoder as drv
    cfg = drv.EncoderConfig(num_layers=3)
    model = drv.HipUniXcoder(drv.synthetic_weights(cfg, 7), cfg, drv.HashTokenizer(cfg.vocab_size), 0)
    rng = np.random.default_rng(5)
    lens = [5, 16, 17, 31, 32, 33, 64, 100, 127, 128, 129, 200, 255, 256, 300, 511, 512, 4, 63, 65]
    rows = [np.concatenate([[0, 6, 2], rng.integers(3, cfg.vocab_size, n - 4), [2]]).astype(np.int32) for n in lens]
    rows = [np.where(r == cfg.pad_token_id, 7, r).astype(np.int32) for r in rows]
    rows[8][40] = cfg.pad_token_id                          # an interior pad token: masked as key, skipped by positions and pool
    L = 512
    padded = np.full((len(rows), L), cfg.pad_token_id, np.int32)
    for i, r in enumerate(rows):
        padded[i, : len(r)] = r
    want = model.forward_ids(torch.from_numpy(padded).to(dev)).cpu().numpy()
    order = list(range(len(rows)))
    flat, off, Lmax = model.pack_rows(rows, order)
    assert Lmax == 512 and off[-1] == sum(lens)
    got = model.forward_packed(torch.from_numpy(flat).to(dev), torch.from_numpy(off).to(dev), Lmax).cpu().numpy()
    assert np.isfinite(got).all()
    assert np.array_equal(got, want), np.abs(got - want).max()       # same arithmetic per token: identical bits
    # twice the same call: identical bits; another row order: every row keeps its vector (to noise)
    again = model.forward_packed(torch.from_numpy(flat).to(dev), torch.from_numpy(off).to(dev), Lmax).cpu().numpy()
    assert np.array_equal(got, again)
    perm = rng.permutation(len(rows)).tolist()
    f2, o2, L2 = model.pack_rows(rows, perm)
    g2 = model.forward_packed(torch.from_numpy(f2).to(dev), torch.from_numpy(o2).to(dev), L2, verify=True).cpu().numpy()
    assert np.array_equal(g2, got[perm])


def test_packed_attention_does_not_touch_its_neighbours(gpu):
    """crh_attn_fwd_packed on ragged rows: every row equals torch softmax attention over ITS tokens only; the rows' last query
    tiles reach into the next row's tokens (computed, never stored), and nothing is written past the last token."""
    torch, ffi, dev = _env()
    H = 12
    lens = [7, 16, 33, 100, 64, 1, 250, 129]
    off = np.zeros(len(lens) + 1, np.int32)
    np.cumsum(lens, out=off[1:])
    T, Lmax = int(off[-1]), 256
    g = torch.Generator(device="cpu").manual_seed(3)
    qkv = torch.randn((T, 3 * H * 64), generator=g).to(dev, torch.bfloat16)
    nw = (Lmax + 63) // 64
    km = torch.zeros((len(lens), nw), dtype=torch.int64)
    for b, n in enumerate(lens):
        bits = [(1 if t < n else 0) for t in range(nw * 64)]
        if b == 3:
            bits[50] = 0                                     # an interior masked key
        for w in range(nw):
            v = sum(bit << i for i, bit in enumerate(bits[w * 64:(w + 1) * 64]))
            km[b, w] = v - (1 << 64) if v >= (1 << 63) else v
    out = torch.full((T + 16, H * 64), 7.0, dtype=torch.bfloat16, device=dev)           # 16 guard rows
    off_d, km_d = torch.from_numpy(off).to(dev), km.to(dev)       # (named: a temporary would be freed -- and its block reused -- before the kernel reads it)
    ffi.check(ffi.lib().crh_attn_fwd_packed(qkv.data_ptr(), off_d.data_ptr(), km_d.data_ptr(), out.data_ptr(), len(lens), T, Lmax, H, 0))
    torch.cuda.synchronize()
    assert bool((out[T:] == 7.0).all())
    for b, n in enumerate(lens):
        x = qkv[off[b]:off[b + 1]].float()
        q, k, v = (t.reshape(n, H, 64).transpose(0, 1) for t in x.split(H * 64, dim=-1))
        s = q @ k.transpose(-1, -2) * 0.125
        if b == 3:
            s[:, :, 50] = float("-inf")
        ref = (torch.softmax(s, -1) @ v).transpose(0, 1).reshape(n, H * 64)
        _close(torch, out[off[b]:off[b + 1]], ref, rel=2 ** -6, abs_=1.5e-2)


def test_embed_texts_pipeline_keeps_order_and_values(gpu, tmp_path):
    """Large embed_texts calls run as a pipeline (tokenizer thread | GPU | float-list conversion through a side stream and
    pinned memory): same vectors, same order as one sequential pass (up to what another batch composition changes)."""
    import glob
    import json
    torch, ffi, dev = _env()
    from safetensors.torch import save_file
    from tokenizers import ByteLevelBPETokenizer
    from coderag_amd import encoder as drv
    d = str(tmp_path)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "code-rag_amd", "*.py")))
    tr = ByteLevelBPETokenizer(add_prefix_space=False)
    tr.train(files, vocab_size=2000, min_frequency=2, special_tokens=["<s>", "<pad>", "</s>", "<unk>", "<mask>", "<encoder-only>"])
    tr.save_model(d)
    cfg = drv.EncoderConfig(vocab_size=2000, num_layers=2)
    json.dump({"vocab_size": 2000, "hidden_size": 768, "num_hidden_layers": 2, "num_attention_heads": 12, "intermediate_size": 3072,
               "max_position_embeddings": 1026, "type_vocab_size": 10}, open(os.path.join(d, "config.json"), "w"))
    save_file({k: torch.from_numpy(v) for k, v in drv.synthetic_weights(cfg, 5).items()}, os.path.join(d, "model.safetensors"))
    model = drv.load_unixcoder(d, device=0)
    src = "".join(open(f, encoding="utf-8").read() for f in files)
    rng = np.random.default_rng(2)
    texts = [src[i:i + int(n)] for i, n in zip(rng.integers(0, len(src) - 900, 3500), rng.integers(5, 900, 3500))]
    model.PIPELINE_CHUNK = 1 << 30
    want = np.asarray(model.embed_texts(texts), np.float32)
    model.PIPELINE_CHUNK = 1000
    try:
        got_list = model.embed_texts(texts)
        got_np = model.embed_texts(texts, rows="numpy")
    finally:
        model.PIPELINE_CHUNK = type(model).PIPELINE_CHUNK
    assert len(got_list) == 3500 and isinstance(got_list[0], list) and isinstance(got_list[0][0], float) and len(got_list[0]) == 768
    got = np.asarray(got_list, np.float32)
    assert np.array_equal(got, np.stack(got_np))
    assert np.abs(got - want).max() <= 2e-2 * np.abs(want).max()
    cos = (got * want).sum(1) / (np.linalg.norm(got, axis=1) * np.linalg.norm(want, axis=1))
    assert cos.min() >= 0.9999, cos.min()
