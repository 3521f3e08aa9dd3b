#!/usr/bin/env python3
"""4096 against 8192 sample tiles behind the thresholds of the int8 scan, on ONE index (the speed of a pass depends on the
allocation, so two indexes would not do), alternating blocks of steps.  python tools/sample_ab.py [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import coderag_amd
from coderag_amd import ffi
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda:0"); D, B, K = 768, 64, 100
st = torch.cuda.current_stream().cuda_stream
os.environ["CODERAG_HIP_I8_SAMPLE"] = "4096"       # what seed_tiles = 4096 (the library's default tuning) selects; 8192 is then asked for by number
idx = ffi.Index(D, ffi.DTYPE_BF16, capacity_rows=rows, device=0)
gen = torch.Generator(device=dev); gen.manual_seed(20251226)
for r0 in range(0, rows, 500_000):
    idx.append(torch.randn((min(500_000, rows - r0), D), generator=gen, device=dev), stream=st)
    torch.cuda.synchronize()
qd = torch.from_numpy(np.random.default_rng(7).standard_normal((B, D)).astype(np.float32)).to(dev)
out = {g: (torch.empty((B, K), dtype=torch.float32, device=dev), torch.empty((B, K), dtype=torch.int64, device=dev)) for g in (4096, 8192)}
times = {4096: [], 8192: []}
cands = {}
for rnd in range(6):
    for g in (4096, 8192):
        idx.set_tuning(seed_tiles=g)
        assert idx.nomination() == ffi.NOMINATE_INT8
        s, r = out[g]
        for _ in range(4):
            idx.search(qd, K, out_scores=s, out_rows=r, stream=st)
        idx.search_finish(st)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
        ev[0].record()
        for i in range(20):
            idx.search(qd, K, out_scores=s, out_rows=r, stream=st)
            ev[i + 1].record()
        idx.search_finish(st); torch.cuda.synchronize()
        times[g] += [ev[i].elapsed_time(ev[i + 1]) for i in range(20)]
        cands[g] = idx.stats()["candidates"] / idx.stats()["batches"] / B
assert torch.equal(out[4096][1], out[8192][1]) and torch.equal(out[4096][0].view(torch.int32), out[8192][0].view(torch.int32))
for g, t in times.items():
    t = np.asarray(t)
    print(f"{g} sample tiles: median {np.median(t):.4f} ms  p10 {np.percentile(t, 10):.4f}  p90 {np.percentile(t, 90):.4f}   candidates per query {cands[g]:.0f}")
