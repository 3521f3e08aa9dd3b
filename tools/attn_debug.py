#!/usr/bin/env python3
"""Where does an attention kernel differ from the fp32 reference?  Prints the max error per (row b, 16-row query tile, head)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import coderag_amd
from coderag_amd import ffi
dev = torch.device("cuda:0"); H = 12
for B, L, full in ((2, 192, True), (2, 192, False), (3, 80, True), (2, 496, True)):
    g = torch.Generator(device="cpu").manual_seed(B * L)
    qkv = torch.randn((B, L, 3 * H * 64), generator=g).to(dev, torch.bfloat16)
    lens = torch.full((B,), L) if full else torch.randint(5, L + 1, (B,), generator=g)
    lens[0] = L
    valid = torch.arange(L)[None, :] < lens[:, None]
    Lp = (L + 63) // 64 * 64
    v = torch.zeros((B, Lp), dtype=torch.bool); v[:, :L] = valid
    km = (v.reshape(B, Lp // 64, 64).to(torch.int64) << torch.arange(64, dtype=torch.int64)).sum(-1).contiguous().to(dev)
    out = torch.full((B, L, H * 64), float("nan"), dtype=torch.bfloat16, device=dev)
    ffi.check(ffi.lib().crh_attn_fwd_varlen(qkv.data_ptr(), km.data_ptr(), out.data_ptr(), B, L, H, 0))
    q, k, vv = (t.reshape(B, L, H, 64).transpose(1, 2).float() for t in qkv.split(H * 64, dim=-1))
    s = (q @ k.transpose(-1, -2) * 0.125).masked_fill(~valid.to(dev)[:, None, None, :], float("-inf"))
    ref = (torch.softmax(s, -1) @ vv).transpose(1, 2).reshape(B, L, H * 64)
    torch.cuda.synchronize()
    err = (out.float() - ref).abs().reshape(B, L, H, 64).amax(-1)           # [B, L, H]
    print(f"B={B} L={L} full={full} lens={lens.tolist()} max err {err.max().item():.4f}")
    for b in range(B):
        nt = (L + 15) // 16
        rows = []
        for t in range(nt):
            e = err[b, t * 16:(t + 1) * 16].amax(0)                        # per head
            rows.append(" ".join("X" if x > 0.05 else ("x" if x > 0.02 else ".") for x in e.tolist()))
        print(f"  b={b}: " + " | ".join(rows))
