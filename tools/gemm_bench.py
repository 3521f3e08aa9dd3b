#!/usr/bin/env python3
"""(variants 100 / 101 = the public entry crh_gemm_bf16_bias with act 0 / 1)
Times the encoder GEMM shapes: the 256x128 3-stage kernel (debug variant 0) against the 256x256 ping-pong kernel
(variant 16), interleaved rounds in one process, random data.  python tools/gemm_bench.py [T]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import coderag_amd
from coderag_amd import ffi
T = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
VARIANTS = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0, 16]
dev = torch.device("cuda:0")
L = ffi.debug_lib()
shapes = [("qkv", 2304, 768), ("oproj", 768, 768), ("ffn1", 3072, 768), ("ffn2", 768, 3072)]
g = torch.Generator(device="cpu").manual_seed(1)
for name, N, K in shapes:
    a = torch.randn((T, K), generator=g).to(dev, torch.bfloat16)
    w = (torch.randn((N, K), generator=g) / K ** 0.5).to(dev, torch.bfloat16)
    b = torch.randn((N,), generator=g).to(dev)
    y = torch.empty((T, N), dtype=torch.bfloat16, device=dev)
    res = torch.randn((T, N), generator=g).to(dev, torch.bfloat16) if N == 768 else None
    gam = torch.ones((768,), device=dev)
    results = {}
    for rnd in range(5):
        for v in VARIANTS:
            if v == 102 and N != 768:
                continue
            def call():
                if v == 102:   # GEMM + residual + LayerNorm (N must be 768)
                    ffi.check(L.crh_gemm_bf16_bias_res_ln(a.data_ptr(), w.data_ptr(), b.data_ptr(), res.data_ptr(), gam.data_ptr(), gam.data_ptr(),
                                                          1e-5, y.data_ptr(), T, N, K, 0))
                elif v >= 100:
                    ffi.check(L.crh_gemm_bf16_bias(a.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), T, N, K, v - 100, 0))
                else:
                    ffi.check(L.crh_debug_gemm_variant(a.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), T, N, K, v, 0))
            for _ in range(3):
                call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                call()
            e1.record()
            torch.cuda.synchronize()
            results.setdefault(v, []).append(e0.elapsed_time(e1) / 20 * 1e3)
    fl = 2.0 * T * N * K
    for v in VARIANTS:
        if v not in results:
            continue
        med = float(np.median(results[v]))
        print(f"{name:6s} T={T} N={N} K={K} variant {v:2d}: median {med:7.1f} us  min {min(results[v]):7.1f}  {fl / med / 1e6:7.0f} TFLOP/s", flush=True)
