#!/usr/bin/env python3
"""From how many rows the int8 nomination copy pays: ms per 64-query top-100 batch (device queries, 40 batches after 10, wall clock
around a synchronize) nominated from the copy (CODERAG_HIP_I8_MIN_ROWS=0) and from the bf16 tiles (set_nomination), same index.
Gaussian rows and the encoder-like anisotropic kind of bench.py.   python tools/i8_crossover.py [rows ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CODERAG_HIP_I8_MIN_ROWS"] = "0"
import numpy as np, torch
import coderag_amd
from coderag_amd import ffi
sizes = [int(a) for a in sys.argv[1:]] or [50_000, 100_000, 200_000, 400_000, 700_000, 1_000_000, 2_000_000]
dev = torch.device("cuda:0"); D, B, K = 768, 64, 100
st = torch.cuda.current_stream().cuda_stream
gen = torch.Generator(device=dev); gen.manual_seed(11)

def rows_of(kind, m):
    g = torch.randn((m, D), generator=gen, device=dev)
    if kind == "anisotropic":
        c = torch.ones((1, D), device=dev) / D ** 0.5
        g = c * 1.68 + g / D ** 0.5          # mean unit vector ~0.86 (bench.py's anisotropic kind, roughly)
    return g

def timed(idx, qd, s, r, steps=40, warm=10):
    for i in range(warm):
        idx.search(qd, K, out_scores=s, out_rows=r, stream=st)
    idx.search_finish(st); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        idx.search(qd, K, out_scores=s, out_rows=r, stream=st)
    idx.search_finish(st); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

for kind in ("gaussian", "anisotropic"):
    for n in sizes:
        idx = ffi.Index(D, ffi.DTYPE_BF16, capacity_rows=n, device=0)
        for r0 in range(0, n, 500_000):
            idx.append(rows_of(kind, min(500_000, n - r0)), stream=st)
        torch.cuda.synchronize()
        qd = rows_of(kind, B)
        s = torch.empty((B, K), dtype=torch.float32, device=dev); r = torch.empty((B, K), dtype=torch.int64, device=dev)
        out = []
        for rnd in range(2):
            idx.set_nomination(ffi.NOMINATE_INT8); a = timed(idx, qd, s, r); fa = idx.stats()["fallback_used"]; ma = idx.nomination()
            idx.set_nomination(ffi.NOMINATE_BF16); b = timed(idx, qd, s, r); mb = idx.nomination()
            out.append((a, b))
        print(f"{kind:12s} {n:>9d} rows: int8 copy {out[0][0]:.4f} / {out[1][0]:.4f} ms (mode {ma}, fallback {fa})   bf16 tiles {out[0][1]:.4f} / {out[1][1]:.4f} ms (mode {mb})", flush=True)
        idx.close()
