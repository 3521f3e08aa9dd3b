#!/usr/bin/env python3
"""Times crh_attn_fwd_varlen on fixed shapes (random data, full-length rows).  python tools/attn_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import coderag_amd
from coderag_amd import ffi
dev = torch.device("cuda:0")
Lb = ffi.lib()
H = 12
for B, L in ((512, 128), (340, 192), (315, 208), (256, 256), (204, 320), (170, 384), (128, 512)):
    qkv = torch.randn((B * L, 3 * H * 64), device=dev).to(torch.bfloat16)
    out = torch.empty((B * L, H * 64), dtype=torch.bfloat16, device=dev)
    nw = (L + 63) // 64
    km = torch.full((B, nw), -1, dtype=torch.int64, device=dev)
    if L % 64:
        km[:, -1] = (1 << (L % 64)) - 1
    ts = []
    for rnd in range(5):
        for _ in range(3):
            ffi.check(Lb.crh_attn_fwd_varlen(qkv.data_ptr(), km.data_ptr(), out.data_ptr(), B, L, H, 0))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ffi.check(Lb.crh_attn_fwd_varlen(qkv.data_ptr(), km.data_ptr(), out.data_ptr(), B, L, H, 0))
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    fl = 4.0 * B * H * L * L * 64
    med = float(np.median(ts))
    print(f"attn B={B} L={L}: median {med:7.1f} us  min {min(ts):7.1f}  {fl / med / 1e6:6.0f} TFLOP/s", flush=True)
