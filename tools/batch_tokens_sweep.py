#!/usr/bin/env python3
"""Tokens per packed batch against encoder throughput on the bench's length mix: the N = 768 GEMMs fill 256 CUs exactly at 85 / 170 /
256 row panels (21 760 / 43 520 / 65 536 tokens), and at the smaller sizes the intermediate tensors (QKV 6 B, FFN hidden 6 KB per
token) fit the 256 MB last-level cache.    python tools/batch_tokens_sweep.py [max_tokens ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import coderag_amd  # noqa: F401
from coderag_amd import encoder as drv

sizes = [int(a) for a in sys.argv[1:]] or [21760, 32768, 43520, 65536, 87040, 131072]
dev = torch.device("cuda:0")
rng = np.random.default_rng(1234)
lengths = np.clip(np.round(np.exp(rng.normal(np.log(160.0), 0.8, 12000))), 8, 512).astype(np.int64)
cfg = drv.EncoderConfig()
m = drv.HipUniXcoder(drv.synthetic_weights(cfg, 23), cfg, drv.HashTokenizer(cfg.vocab_size), 0)
rows = [np.concatenate([[0, 5, 2], rng.integers(16, cfg.vocab_size, int(L) - 4), [2]]).astype(np.int32) for L in lengths]
res = {}
plans = {}
for mt in sizes:
    plan = m.plan_batches([len(r) for r in rows], max_tokens=mt, max_rows=4096, packed=True)
    batches = []
    for idx, _ in plan:
        flat, off, Lmax = m.pack_rows(rows, list(idx))
        batches.append((torch.from_numpy(flat).to(dev), torch.from_numpy(off).to(dev), Lmax, len(flat)))
    plans[mt] = batches
for rnd in range(4):
    for mt in sizes:
        batches = plans[mt]
        for i, o, L, _ in batches[:2]:
            m.forward_packed(i, o, L)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i, o, L, _ in batches:
            m.forward_packed(i, o, L)
        torch.cuda.synchronize()
        res.setdefault(mt, []).append(sum(b[3] for b in batches) / (time.perf_counter() - t0))
for mt in sizes:
    print(f"max_tokens {mt:>7d}: {len(plans[mt]):3d} batches, tokens/s by round {[round(v / 1e6, 3) for v in res[mt]]} M  median {np.median(res[mt]) / 1e6:.3f} M")
