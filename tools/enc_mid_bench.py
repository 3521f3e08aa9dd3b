#!/usr/bin/env python3
"""Encoder forward over a grid of batch shapes, from one query to 64k tokens -- the data behind the GEMM kernel choice
(k_gemm_mid / k_gemm_nt / ping-pong, crh_encoder.hip choose_gemm).  Run once per dispatch setting, e.g.
  CODERAG_HIP_GEMM256=2 python tools/enc_mid_bench.py        (ping-pong kernel whatever the tile count)
  CODERAG_HIP_MID=0 python tools/enc_mid_bench.py           (no k_gemm_mid)
Shapes as arguments (BxL, e.g. 16x128) replace the default sweep.  Prints device ms per forward (median of 30, torch events) and chunks/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import coderag_amd  # noqa: F401
from coderag_amd import encoder as drv

cfg = drv.EncoderConfig()
model = drv.HipUniXcoder(drv.synthetic_weights(cfg, 23), cfg, drv.HashTokenizer(cfg.vocab_size), 0)
dev = torch.device("cuda:0")
rng = np.random.default_rng(1)
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("CODERAG_HIP_")) or "default"
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(8, 128), (16, 128), (32, 128), (64, 128), (128, 128), (256, 128), (4, 512), (8, 512), (16, 512), (32, 512), (24, 160)]
for B, L in shapes:
    ids = torch.from_numpy(rng.integers(16, cfg.vocab_size, (B, L)).astype(np.int32)).to(dev)
    for _ in range(5):
        model.forward_ids(ids)
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        model.forward_ids(ids)
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    ms = float(np.median(ts))
    flops = B * (169_869_312 * L + 36_864 * L * L)
    print(f"[{tag}] B={B:4d} L={L:4d} T={B * L:6d}: {ms:7.3f} ms  {B / ms * 1e3:8.0f} chunks/s  {flops / ms / 1e9:6.0f} TFLOP/s", flush=True)
