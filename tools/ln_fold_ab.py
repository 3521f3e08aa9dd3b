#!/usr/bin/env python3
"""LayerNorm folded into the GEMMs around it (EncoderConfig.ln_fold, round 5) against the two LayerNorm kernels per layer of
rounds 1-4: the packed forward on the bench's length mix (lognormal lengths, ~65 k tokens per batch), alternating rounds on one
device; ms per batch (median of the rounds) and the per-kernel split comes from a kernel trace of the same script.

    python tools/ln_fold_ab.py [batches per round] [rounds]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import coderag_amd  # noqa: F401
from coderag_amd import encoder as drv

nb, rounds = (int(sys.argv[1]) if len(sys.argv) > 1 else 12), (int(sys.argv[2]) if len(sys.argv) > 2 else 5)
dev = torch.device("cuda:0")
rng = np.random.default_rng(1234)
lengths = np.clip(np.round(np.exp(rng.normal(np.log(160.0), 0.8, 4000))), 8, 512).astype(np.int64)
models = {}
for fold in (False, True, "residual_f32"):
    cfg = drv.EncoderConfig(residual_f32=True) if fold == "residual_f32" else drv.EncoderConfig(ln_fold=fold)
    models[fold] = drv.HipUniXcoder(drv.synthetic_weights(cfg, 23), cfg, drv.HashTokenizer(cfg.vocab_size), 0)
m0 = models[False]
rows = [np.concatenate([[0, 5, 2], rng.integers(16, m0.cfg.vocab_size, int(L) - 4), [2]]).astype(np.int32) if L >= 4 else np.array([0, 5, 2, 2], np.int32) for L in lengths]
plan = m0.plan_batches([len(r) for r in rows], packed=True)[:nb]
batches = []
for b in plan:
    idx = list(b[0]) if isinstance(b, tuple) else list(b)
    flat, off, Lmax = m0.pack_rows(rows, idx)
    batches.append((torch.from_numpy(flat).to(dev), torch.from_numpy(off).to(dev), Lmax, len(flat)))
tok = sum(b[3] for b in batches)
out = {}
for fold, m in models.items():
    out[fold] = [m.forward_packed(i, o, L).clone() for i, o, L, _ in batches[:2]]
torch.cuda.synchronize()
cos = [float(torch.nn.functional.cosine_similarity(a, b).min()) for a, b in zip(out[True], out[False])]
print(f"{len(batches)} packed batches, {tok} tokens; folded vs unfolded embeddings: min cosine {min(cos):.6f}")
times = {k: [] for k in models}
for r in range(rounds):
    for fold in models:
        m = models[fold]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i, o, L, _ in batches:
            m.forward_packed(i, o, L)
        torch.cuda.synchronize()
        times[fold].append((time.perf_counter() - t0) * 1e3 / len(batches))
for fold in models:
    print(f"ln_fold={fold}: ms per batch by round {[round(t, 3) for t in times[fold]]}  median {np.median(times[fold]):.3f}")
a, b = np.median(times[False]), np.median(times[True])
print(f"folded / unfolded = {b / a:.4f}  ({(1 - b / a) * 100:+.2f} % time)")
c = np.median(times["residual_f32"])
print(f"residual_f32 / default = {c / a:.4f}")
