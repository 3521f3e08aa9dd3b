#!/usr/bin/env python3
"""Encoder latency on the query path (one short text): forward_ids wall time incl. the final sync, median of 50.
CODERAG_HIP_SKINNY=0 shows the tiled kernels on the same shapes.  python tools/latency_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import coderag_amd
from coderag_amd import encoder as drv
cfg = drv.EncoderConfig()
model = drv.HipUniXcoder(drv.synthetic_weights(cfg, 23), cfg, drv.HashTokenizer(cfg.vocab_size), 0)
dev = torch.device("cuda:0")
rng = np.random.default_rng(1)
for B, L in ((1, 16), (1, 64), (1, 256), (1, 384), (1, 512), (8, 64), (16, 64), (32, 64), (4, 512)):
    ids = rng.integers(16, cfg.vocab_size, (B, L)).astype(np.int32)
    t = torch.from_numpy(ids).to(dev)
    for name, fn in (("forward", model.forward_ids),):
        for _ in range(5):
            fn(t)
        torch.cuda.synchronize()
        ts = []
        for _ in range(50):
            t0 = time.perf_counter()
            fn(t)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        print(f"B={B} L={L} {name}: median {np.median(ts):.3f} ms  p90 {np.percentile(ts, 90):.3f}", flush=True)
