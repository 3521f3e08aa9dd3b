#!/usr/bin/env python3
"""Reference point for the encoder GEMMs: the same shapes through torch (hipBLASLt / rocBLAS), bias only, random data."""
import numpy as np, torch
dev = torch.device("cuda:0")
T = 65536
for name, N, K in (("qkv", 2304, 768), ("oproj", 768, 768), ("ffn1", 3072, 768), ("ffn2", 768, 3072)):
    a = torch.randn((T, K), device=dev).to(torch.bfloat16)
    w = (torch.randn((N, K), device=dev) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn((N,), device=dev).to(torch.bfloat16)
    ts = []
    for rnd in range(5):
        for _ in range(3):
            torch.nn.functional.linear(a, w, b)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            torch.nn.functional.linear(a, w, b)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    med = float(np.median(ts))
    print(f"{name:6s} torch.linear bf16 T={T} N={N} K={K}: median {med:7.1f} us  {2.0 * T * N * K / med / 1e6:7.0f} TFLOP/s", flush=True)
