#!/usr/bin/env python3
"""HBM read ceiling of the scan's access pattern (crh_debug_read_ceiling) beside the real scan, same process, interleaved.
python tools/read_ceiling.py [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
# the probe entry point exists in the debug build only: this tool's whole run (index, scan, probe) uses that library
os.environ.setdefault("CODERAG_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "code-rag_amd", "lib", "libcoderag_hip_debug.so"))
import coderag_amd
from coderag_amd import ffi
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda:0")
idx = ffi.Index(768, ffi.DTYPE_BF16, capacity_rows=rows)
gen = torch.Generator(device=dev); gen.manual_seed(1)
for r0 in range(0, rows, 500_000):
    idx.append(torch.randn((min(500_000, rows - r0), 768), generator=gen, device=dev))
torch.cuda.synchronize()
q = torch.randn((64, 768), generator=gen, device=dev)
s = torch.empty((64, 100), dtype=torch.float32, device=dev); r = torch.empty((64, 100), dtype=torch.int64, device=dev)
L = ffi.lib()
st = torch.cuda.current_stream().cuda_stream
res = {"probe": [], "scan": []}
for rnd in range(5):
    for _ in range(3):
        ffi.check(L.crh_debug_read_ceiling(idx._handle(), st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ffi.check(L.crh_debug_read_ceiling(idx._handle(), st))
    e1.record(); torch.cuda.synchronize()
    res["probe"].append(e0.elapsed_time(e1) / 20)
    idx.set_profiling(True)
    for _ in range(20):
        idx.search(q, 100, out_scores=s, out_rows=r, stream=st)
    idx.search_finish(st); torch.cuda.synchronize()
    ms, n = idx.profile(); idx.set_profiling(False)
    res["scan"].append(ms / n)
gb = rows * 768 * 2 / 1e9
for k, v in res.items():
    m = float(np.median(v))
    print(f"{k}: median {m:.3f} ms = {gb / m:.2f} TB/s", flush=True)
