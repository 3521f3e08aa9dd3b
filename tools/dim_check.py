#!/usr/bin/env python3
"""One width at size: N rows x D, 64 queries, top-100 -- nominated from the int8 copy against the bf16 tiles on the same index:
identical ids and score bits, ms per batch of both (40 batches after 10).   python tools/dim_check.py [D] [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import coderag_amd
from coderag_amd import ffi
D = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
dev = torch.device("cuda:0"); B, K = (32 if D == 1536 else 64), 100
st = torch.cuda.current_stream().cuda_stream
gen = torch.Generator(device=dev); gen.manual_seed(5)
idx = ffi.Index(D, ffi.DTYPE_BF16, capacity_rows=n, device=0)
for r0 in range(0, n, 500_000):
    idx.append(torch.randn((min(500_000, n - r0), D), generator=gen, device=dev), stream=st)
torch.cuda.synchronize()
qd = torch.randn((B, D), generator=gen, device=dev)
res = {}
for mode, name in ((ffi.NOMINATE_INT8, "int8 copy"), (ffi.NOMINATE_BF16, "bf16 tiles")):
    idx.set_nomination(mode)
    s = torch.empty((B, K), dtype=torch.float32, device=dev); r = torch.empty((B, K), dtype=torch.int64, device=dev)
    for i in range(10):
        idx.search(qd, K, out_scores=s, out_rows=r, stream=st)
    idx.search_finish(st); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(40):
        idx.search(qd, K, out_scores=s, out_rows=r, stream=st)
    idx.search_finish(st); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 40 * 1e3
    res[name] = (s.clone(), r.clone())
    print(f"dim {D}, {n} rows, batch {B}: {name:10s} {ms:.4f} ms per batch, mode {idx.nomination()}, stats {idx.stats()['fallback_used']}", flush=True)
a, b = res["int8 copy"], res["bf16 tiles"]
print("identical ids:", bool(torch.equal(a[1], b[1])), " identical score bits:", bool(torch.equal(a[0].view(torch.int32), b[0].view(torch.int32))))
