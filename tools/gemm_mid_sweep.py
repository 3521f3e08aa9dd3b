#!/usr/bin/env python3
"""Per-GEMM timing through the public entry over a grid of T for the encoder's four shapes -- run once per dispatch
setting (CODERAG_HIP_MID=0|2, CODERAG_HIP_GEMM256=0|2; unset = the cost model) and compare.  us per call, median of 40."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import coderag_amd  # noqa: F401
from coderag_amd import ffi

dev = torch.device("cuda:0")
Ts = [int(a) for a in sys.argv[1:]] or [640, 1024, 1536, 2048, 2560, 3072, 4096, 5120, 6144, 7168, 8192, 10240, 12288, 16384, 20480]
shapes = [("qkv", 2304, 768, 0), ("oproj", 768, 768, 0), ("ffn1", 3072, 768, 1), ("ffn2", 768, 3072, 0)]
L = ffi.lib()
for T in Ts:
    out = []
    for name, N, K, act in shapes:
        a = torch.randn((T, K), device=dev).to(torch.bfloat16)
        w = (torch.randn((N, K), device=dev) / K ** 0.5).to(torch.bfloat16)
        b = torch.randn((N,), device=dev)
        y = torch.empty((T, N), dtype=torch.bfloat16, device=dev)
        for _ in range(5):
            ffi.check(L.crh_gemm_bf16_bias(a.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), T, N, K, act, 0))
        ts = []
        for _ in range(40):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ffi.check(L.crh_gemm_bf16_bias(a.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), T, N, K, act, 0))
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        out.append(f"{name} {np.median(ts):7.1f}")
    print(f"T={T:6d}  " + "  ".join(out), flush=True)
