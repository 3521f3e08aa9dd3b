#!/usr/bin/env python3
"""k_scan_fused (seed scan + threshold + main scan in one launch) against the three-launch form, ONE process, two indexes over the
same 10M x 768 bf16 corpus, alternating blocks of steps: per-step device time (events on the launch stream), medians.
python tools/fused_ab.py [rows] [swap]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import coderag_amd
from coderag_amd import ffi
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda:0"); D, B, K = 768, 64, 100
st = torch.cuda.current_stream().cuda_stream


def build(fused):
    os.environ["CODERAG_HIP_FUSED_SCAN"] = "1" if fused else "0"
    idx = ffi.Index(D, ffi.DTYPE_BF16, capacity_rows=rows, device=0)
    gen = torch.Generator(device=dev); gen.manual_seed(20251226)
    for r0 in range(0, rows, 500_000):
        m = min(500_000, rows - r0)
        idx.append(torch.randn((m, D), generator=gen, device=dev), stream=st)
        torch.cuda.synchronize()
    return idx


order = ("three", "fused") if len(sys.argv) > 2 and sys.argv[2] == "swap" else ("fused", "three")
idx = {name: build(name == "fused") for name in order}
qd = torch.from_numpy(np.random.default_rng(7).standard_normal((B, D)).astype(np.float32)).to(dev)
out = {k: (torch.empty((B, K), dtype=torch.float32, device=dev), torch.empty((B, K), dtype=torch.int64, device=dev)) for k in idx}
times = {k: [] for k in idx}
for rnd in range(8):
    for name, ix in idx.items():
        s, r = out[name]
        for _ in range(3):
            ix.search(qd, K, out_scores=s, out_rows=r, stream=st)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
        ev[0].record()
        for i in range(20):
            ix.search(qd, K, out_scores=s, out_rows=r, stream=st)
            ev[i + 1].record()
        ix.search_finish(st)
        torch.cuda.synchronize()
        times[name] += [ev[i].elapsed_time(ev[i + 1]) for i in range(20)]
assert torch.equal(out["fused"][1], out["three"][1]) and torch.equal(out["fused"][0].view(torch.int32), out["three"][0].view(torch.int32))
for name, t in times.items():
    t = np.asarray(t)
    print(f"{name}: median {np.median(t):.4f} ms  p10 {np.percentile(t, 10):.4f}  p90 {np.percentile(t, 90):.4f}  mean {t.mean():.4f}   ({len(t)} steps)")
print(f"difference of medians: {(np.median(times['three']) - np.median(times['fused'])) * 1e3:.1f} us in favour of the fused form")
