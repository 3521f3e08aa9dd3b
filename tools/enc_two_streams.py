#!/usr/bin/env python3
"""Experiment: the encoder's packed batches on ONE stream vs alternating between TWO (one batch's kernel tails and launch gaps
filled by the other batch's kernels).  python tools/enc_two_streams.py [chunks]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import coderag_amd
from coderag_amd import encoder as drv
n_chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
cfg = drv.EncoderConfig()
model = drv.HipUniXcoder(drv.synthetic_weights(cfg, 23), cfg, drv.HashTokenizer(cfg.vocab_size), 0)
rng = np.random.default_rng(1234)
lengths = np.clip(np.round(np.exp(rng.normal(np.log(160.0), 0.8, n_chunks))), 8, 512).astype(np.int64)
dev = torch.device("cuda", 0)
id_rows = [np.concatenate([[0, 5, 2], rng.integers(16, cfg.vocab_size, int(n) - 4), [2]]).astype(np.int32) if n >= 4
           else np.asarray([0, 5, 2, 2][:int(n)], np.int32) for n in lengths]
for max_tokens in (65536, 32768):
    batches = []
    for rows, _ in model.plan_batches(lengths, max_tokens=max_tokens, max_rows=4096, packed=True):
        flat, off, Lmax = model.pack_rows(id_rows, rows)
        batches.append((torch.from_numpy(flat).to(dev), torch.from_numpy(off).to(dev), Lmax))
    flops = float(sum(drv.flops_per_chunk(int(n), cfg) for n in lengths))
    side = [torch.cuda.Stream(device=dev) for _ in range(3)]
    for ns in (1, 2, 3, 1, 2):
        outs = []
        for ids, off, Lmax in batches[:3]:
            model.forward_packed(ids, off, Lmax)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if ns == 1:
            for ids, off, Lmax in batches:
                outs.append(model.forward_packed(ids, off, Lmax))
        else:
            cur = torch.cuda.current_stream(dev)
            for s in side[:ns]:
                s.wait_stream(cur)
            for i, (ids, off, Lmax) in enumerate(batches):
                with torch.cuda.stream(side[i % ns]):
                    outs.append(model.forward_packed(ids, off, Lmax))
            for s in side[:ns]:
                cur.wait_stream(s)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"max_tokens {max_tokens} streams {ns}: {n_chunks / dt:8.0f} chunks/s  {flops / dt / 1e12:6.0f} TFLOP/s  ({len(batches)} batches)", flush=True)
