#!/usr/bin/env python3
"""Does the speed of the int8 scan depend on WHICH allocation holds the copy?  One process, several 10M x 768 indexes built one
after the other (all alive), the headline batch timed on each in turn, twice over.  python tools/alloc_modes.py [indexes] [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import coderag_amd
from coderag_amd import ffi
nidx = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
dev = torch.device("cuda:0"); D, B, K = 768, 64, 100
st = torch.cuda.current_stream().cuda_stream
qd = torch.from_numpy(np.random.default_rng(7).standard_normal((B, D)).astype(np.float32)).to(dev)
s = torch.empty((B, K), dtype=torch.float32, device=dev); r = torch.empty((B, K), dtype=torch.int64, device=dev)


def build():
    idx = ffi.Index(D, ffi.DTYPE_BF16, capacity_rows=rows, device=0)
    gen = torch.Generator(device=dev); gen.manual_seed(20251226)
    for r0 in range(0, rows, 500_000):
        idx.append(torch.randn((min(500_000, rows - r0), D), generator=gen, device=dev), stream=st)
        torch.cuda.synchronize()
    return idx


def timed(idx, mode, n=30):
    idx.set_nomination(mode)
    for _ in range(5):
        idx.search(qd, K, out_scores=s, out_rows=r, stream=st)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        idx.search(qd, K, out_scores=s, out_rows=r, stream=st)
        ev[i + 1].record()
    idx.search_finish(st); torch.cuda.synchronize()
    return float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(n)]))


idxs = []
for i in range(nidx):
    idxs.append(build())
    print(f"index {i} built", flush=True)
for rnd in range(2):
    for i, idx in enumerate(idxs):
        print(f"round {rnd} index {i}: int8 {timed(idx, ffi.NOMINATE_INT8):.4f} ms   bf16 {timed(idx, ffi.NOMINATE_BF16):.4f} ms", flush=True)
