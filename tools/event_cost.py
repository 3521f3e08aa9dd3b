#!/usr/bin/env python3
"""What the bench's own instrumentation costs per batch: 10M x 768 bf16 rows, batch 64, top-100, 50 steps after 10 warm-up steps,
wall clock around a full synchronize -- (a) nothing recorded, (b) the library's two HIP events around the pass (set_profiling),
(c) one torch event per step on the launch stream, (d) both (what bench.py's timed region does).  Interleaved, three rounds.
python tools/event_cost.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import coderag_amd
from coderag_amd import ffi
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda:0"); D, B, K = 768, 64, 100
st = torch.cuda.current_stream().cuda_stream
idx = ffi.Index(D, ffi.DTYPE_BF16, capacity_rows=rows, device=0)
gen = torch.Generator(device=dev); gen.manual_seed(20251226)
for r0 in range(0, rows, 500_000):
    m = min(500_000, rows - r0)
    idx.append(torch.randn((m, D), generator=gen, device=dev), stream=st)
    torch.cuda.synchronize()
qd = torch.randn((B, D), generator=gen, device=dev)
s = [torch.empty((B, K), dtype=torch.float32, device=dev) for _ in range(4)]
r = [torch.empty((B, K), dtype=torch.int64, device=dev) for _ in range(4)]

def run(prof, step_events, steps=50, warm=10):
    idx.set_profiling(prof)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    for i in range(warm):
        idx.search(qd, K, out_scores=s[i % 4], out_rows=r[i % 4], stream=st)
    idx.search_finish(st); torch.cuda.synchronize()
    t0 = time.perf_counter()
    if step_events: ev[0].record()
    for i in range(steps):
        idx.search(qd, K, out_scores=s[i % 4], out_rows=r[i % 4], stream=st)
        if step_events: ev[i + 1].record()
    idx.search_finish(st); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    idx.set_profiling(False)
    return dt

for rnd in range(3):
    out = []
    for prof, se in ((False, False), (True, False), (False, True), (True, True)):
        out.append(run(prof, se))
    print("round %d: plain %.4f  pass events %.4f  step events %.4f  both %.4f ms per batch" % (rnd, *out), flush=True)
