#!/usr/bin/env python3
"""Does the streaming scan need every CU?  k_scan time per pass on 10M x 768 bf16 with the grid cut by CODERAG_HIP_SPARE_CUS."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import coderag_amd
from coderag_amd import ffi
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda:0"); D, K = 768, 100
qs = torch.from_numpy(np.random.default_rng(7).standard_normal((64, D)).astype(np.float32)).to(dev)
st = torch.cuda.current_stream().cuda_stream
for spare in (0, 8, 16, 32):
    os.environ["CODERAG_HIP_SPARE_CUS"] = str(spare)
    idx = ffi.Index(D, ffi.DTYPE_BF16, capacity_rows=rows)
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    for r0 in range(0, rows, 500_000):
        idx.append(torch.randn((min(500_000, rows - r0), D), generator=gen, device=dev)); torch.cuda.synchronize()
    s = torch.empty((64, K), dtype=torch.float32, device=dev); r = torch.empty((64, K), dtype=torch.int64, device=dev)
    for _ in range(3): idx.search(qs, K, out_scores=s, out_rows=r, stream=st)
    idx.search_finish(st); torch.cuda.synchronize(); idx.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(20): idx.search(qs, K, out_scores=s, out_rows=r, stream=st)
    idx.search_finish(st); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    ms, n = idx.profile()
    print(json.dumps({"spare_cus": spare, "ms_per_call": dt * 1e3, "scan_ms": ms / n}), flush=True)
    idx.close(); del idx
