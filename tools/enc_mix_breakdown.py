#!/usr/bin/env python3
"""Where the encoder leg's time goes on the bench mix, by padded length bucket, and how the batch size (max_tokens) moves it.
python tools/enc_mix_breakdown.py [n_chunks]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import coderag_amd
from coderag_amd import encoder as drv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
cfg = drv.EncoderConfig()
model = drv.HipUniXcoder(drv.synthetic_weights(cfg, 23), cfg, drv.HashTokenizer(cfg.vocab_size), 0)
rng = np.random.default_rng(1234)
lengths = np.clip(np.round(np.exp(rng.normal(np.log(160.0), 0.8, n))), 8, 512).astype(np.int64)
dev = torch.device("cuda:0")
for max_tokens in (32768, 65536, 98304, 131072):
    batches, meta = [], []
    for rows, L in model.plan_batches(lengths, max_tokens=max_tokens, max_rows=4096 if max_tokens > 65536 else 1024):
        host = rng.integers(16, cfg.vocab_size, (len(rows), L)).astype(np.int32)
        for r, i in enumerate(rows):
            host[r, int(lengths[i]):] = cfg.pad_token_id
        batches.append(torch.from_numpy(host).to(dev))
        meta.append((L, len(rows), float(sum(drv.flops_per_chunk(int(lengths[i]), cfg) for i in rows))))
    for b in batches[:3]:
        model.forward_ids(b)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(batches) + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i, b in enumerate(batches):
        model.forward_ids(b)
        ev[i + 1].record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(len(batches))])
    fl = np.array([m[2] for m in meta]); Ls = np.array([m[0] for m in meta]); toks = np.array([m[0] * m[1] for m in meta])
    out = {"max_tokens": max_tokens, "batches": len(batches), "chunks_per_s": n / dt, "TFLOPs": fl.sum() / dt / 1e12, "buckets": []}
    for lo, hi in ((0, 64), (64, 128), (128, 192), (192, 256), (256, 384), (384, 513)):
        m = (Ls > lo) & (Ls <= hi)
        if m.any():
            out["buckets"].append({"L": f"{lo + 1}-{hi}", "batches": int(m.sum()), "ms": float(ms[m].sum()), "share": float(ms[m].sum() / ms.sum()),
                                   "TFLOPs": float(fl[m].sum() / ms[m].sum() / 1e9), "mean_tokens_per_batch": float(toks[m].mean())})
    print(json.dumps(out), flush=True)
