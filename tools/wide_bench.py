#!/usr/bin/env python3
"""Large query batches: queries/s of one search call with nq in {64, 128, 256, 512, 1024} on a 10M x 768 bf16 corpus --
k_scan_wide (one corpus pass per 256 queries) against the 64-query passes (CODERAG_HIP_NO_WIDE_SCAN=1).
python tools/wide_bench.py [rows] -> JSON lines."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import coderag_amd
from coderag_amd import ffi
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda:0")
D, K = 768, 100


def build():
    idx = ffi.Index(D, ffi.DTYPE_BF16, capacity_rows=rows)
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    for r0 in range(0, rows, 500_000):
        idx.append(torch.randn((min(500_000, rows - r0), D), generator=gen, device=dev))
        torch.cuda.synchronize()
    return idx

qs = torch.from_numpy(np.random.default_rng(7).standard_normal((1024, D)).astype(np.float32)).to(dev)
st = torch.cuda.current_stream().cuda_stream
res = {}
for mode in ("wide", "narrow"):
    if mode == "narrow":
        os.environ["CODERAG_HIP_NO_WIDE_SCAN"] = "1"
    idx = build()
    os.environ.pop("CODERAG_HIP_NO_WIDE_SCAN", None)
    for nq in (64, 65, 128, 256, 512, 1024):
        s = torch.empty((nq, K), dtype=torch.float32, device=dev); r = torch.empty((nq, K), dtype=torch.int64, device=dev)
        for _ in range(3):
            idx.search(qs[:nq], K, out_scores=s, out_rows=r, stream=st)
        idx.search_finish(st); torch.cuda.synchronize()
        idx.set_profiling(True)
        n = 10
        t0 = time.perf_counter()
        for _ in range(n):
            idx.search(qs[:nq], K, out_scores=s, out_rows=r, stream=st)
        idx.search_finish(st); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        ms, launches = idx.profile(); idx.set_profiling(False)
        stats = idx.stats()
        res[(mode, nq)] = r.cpu().numpy()
        same = bool(np.array_equal(res[("wide", nq)], res[(mode, nq)]))
        print(json.dumps({"mode": mode, "nq": nq, "ms_per_call": dt * 1e3, "queries_per_s": nq / dt, "scan_ms_per_pass": ms / max(1, launches),
                          "passes_per_call": launches / n, "scan_GBps": rows * D * 2 / (ms / max(1, launches) * 1e-3) / 1e9,
                          "mfma_TFLOPs_in_scan": 2.0 * min(nq, 256) * rows * D / (ms / max(1, launches) * 1e-3) / 1e12 if mode == "wide" and nq > 64 else None,
                          "candidates": stats["candidates"], "fallback": stats["fallback_used"], "ids_equal_to_wide": same}), flush=True)
    idx.close()
