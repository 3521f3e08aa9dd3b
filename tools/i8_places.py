#!/usr/bin/env python3
"""The headline batch on ONE index whose int8 copy is moved to a fresh allocation between rounds (crh_debug_i8_move, debug
build): does the place of the copy decide the speed of the pass?  python tools/i8_places.py [rounds] [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CODERAG_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "code-rag_amd", "lib", "libcoderag_hip_debug.so"))
import numpy as np, torch
import coderag_amd
from coderag_amd import ffi
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
dev = torch.device("cuda:0"); D, B, K = 768, 64, 100
st = torch.cuda.current_stream().cuda_stream
idx = ffi.Index(D, ffi.DTYPE_BF16, capacity_rows=rows, device=0)
gen = torch.Generator(device=dev); gen.manual_seed(20251226)
for r0 in range(0, rows, 500_000):
    idx.append(torch.randn((min(500_000, rows - r0), D), generator=gen, device=dev), stream=st)
    torch.cuda.synchronize()
qd = torch.from_numpy(np.random.default_rng(7).standard_normal((B, D)).astype(np.float32)).to(dev)
s = torch.empty((B, K), dtype=torch.float32, device=dev); r = torch.empty((B, K), dtype=torch.int64, device=dev)
L = ffi.lib()


def timed(mode, n=30):
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


def both(tag):
    print(f"{tag}: int8 {timed(ffi.NOMINATE_INT8):.4f} ms   bf16 tiles {timed(ffi.NOMINATE_BF16):.4f} ms", flush=True)
    idx.set_nomination(ffi.NOMINATE_INT8)


both("as built")
both("again")
x = torch.empty(8 << 30, dtype=torch.uint8, device=dev); torch.cuda.synchronize()
both("8 GB more allocated (torch)")
del x; torch.cuda.synchronize()
both("... and dropped (cached by torch)")
torch.cuda.empty_cache(); torch.cuda.synchronize()
both("... and released (hipFree)")
for rnd in range(rounds):
    ffi.check(L.crh_debug_i8_move(idx._handle()), L)
    both(f"copy moved {rnd + 1}x")
