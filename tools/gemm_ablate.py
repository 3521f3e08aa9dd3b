#!/usr/bin/env python3
"""A/B timing of the GEMM main-loop ablations in one process (interleaved rounds, HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import coderag_amd  # noqa: F401
from coderag_amd import ffi

L = ffi.debug_lib()
dev = torch.device("cuda:0")
for (T, N, K) in ((32768, 2304, 768), (32768, 768, 3072)):
    a = torch.randn((T, K), device=dev).to(torch.bfloat16)
    w = (torch.randn((N, K), device=dev) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn((N,), device=dev)
    y = torch.empty((T, N), dtype=torch.bfloat16, device=dev)
    VARS = (0, 1, 5, 2, 6, 8)
    res = {v: [] for v in VARS}
    for rnd in range(6):
        for v in VARS:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ffi.check(L.crh_debug_gemm_variant(a.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), T, N, K, v, 0))
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                res[v].append(e0.elapsed_time(e1) / 5 * 1e3)
    fl = 2.0 * T * N * K
    names = {0: "full", 1: "no DMA in loop", 2: "no MFMA", 4: "no fragment reads", 5: "MFMA only (no DMA, no frag reads)",
             6: "DMA only (no MFMA, no frag reads)", 8: "no W DMA"}
    for v in VARS:
        us = sorted(res[v])[len(res[v]) // 2]
        print(f"T={T} N={N} K={K} {names[v]:34s} median {us:8.1f} us  ({fl / us / 1e6:7.1f} TFLOP/s-equivalent)")
