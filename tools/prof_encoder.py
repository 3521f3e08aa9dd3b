#!/usr/bin/env python3
"""Fixed-shape encoder runs for rocprofv3 (kernel breakdown of one forward).  Usage: prof_encoder.py B L [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import coderag_amd  # noqa: F401
from coderag_amd import encoder as drv

B, L = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
cfg = drv.EncoderConfig()
model = drv.HipUniXcoder(drv.synthetic_weights(cfg, 23), cfg, drv.HashTokenizer(cfg.vocab_size), 0)
rng = np.random.default_rng(0)
ids = torch.from_numpy(rng.integers(16, cfg.vocab_size, (B, L)).astype(np.int32)).cuda()
model.forward_ids(ids)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    model.forward_ids(ids)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
fl = B * drv.flops_per_chunk(L, cfg)
print(f"B={B} L={L}: {dt * 1e3:.3f} ms/forward, {B / dt:.0f} chunks/s, {fl / dt / 1e12:.1f} TFLOP/s")
