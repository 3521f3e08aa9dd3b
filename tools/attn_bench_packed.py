#!/usr/bin/env python3
"""crh_attn_fwd_packed on ~65k tokens of equal-length packed rows (the encoder's batches).  python tools/attn_bench_packed.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import coderag_amd
from coderag_amd import ffi
dev = torch.device("cuda:0"); Lb = ffi.lib(); H = 12
for L in (40, 72, 100, 128, 150, 192, 208, 250, 256, 300, 320, 384, 450, 512):
    B = max(1, 65536 // L)
    T = B * L
    qkv = torch.randn((T, 3 * H * 64), device=dev).to(torch.bfloat16)
    out = torch.empty((T, H * 64), dtype=torch.bfloat16, device=dev)
    Lmax = (L + 15) // 16 * 16
    nw = (Lmax + 63) // 64
    km = torch.zeros((B, nw), dtype=torch.int64)
    for w in range(nw):
        bits = min(64, max(0, L - 64 * w))
        km[:, w] = -1 if bits == 64 else (1 << bits) - 1
    km = km.to(dev)
    off = torch.arange(0, T + 1, L, dtype=torch.int32, device=dev)
    ts = []
    for rnd in range(5):
        for _ in range(3):
            ffi.check(Lb.crh_attn_fwd_packed(qkv.data_ptr(), off.data_ptr(), km.data_ptr(), out.data_ptr(), B, T, Lmax, H, 0))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ffi.check(Lb.crh_attn_fwd_packed(qkv.data_ptr(), off.data_ptr(), km.data_ptr(), out.data_ptr(), B, T, Lmax, H, 0))
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    fl = 4.0 * B * H * L * L * 64
    med = float(np.median(ts))
    print(f"attn packed B={B} L={L}: median {med:7.1f} us  {fl / med / 1e6:6.0f} TFLOP/s", flush=True)
