#!/usr/bin/env python3
"""Where k_scan_fused spends its time before the main loop: per-workgroup wall-clock stamps (100 MHz) taken by a measurement
build of the library (tools/build_stamps_lib.sh: -DCRH_FUSED_STAMPS -> lib/libcoderag_hip_stamps.so; not part of build.sh):
0 kernel entry, 1 query image in LDS, 2 sample tile done, 3 past wait A, 4 threshold written, 5 past wait B, 6 main loop done.
CODERAG_HIP_LIB=code-rag_amd/lib/libcoderag_hip_stamps.so python tools/fused_stamps.py [rows]"""
import ctypes as C, os, sys
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
s = torch.empty((B, K), dtype=torch.float32, device=dev); r = torch.empty((B, K), dtype=torch.int64, device=dev)
L = ffi.lib()
L.crh_debug_fused_stamps.argtypes = [C.c_void_p]; L.crh_debug_fused_stamps.restype = C.c_int
L.crh_debug_select_stamps.argtypes = [C.c_void_p]; L.crh_debug_select_stamps.restype = C.c_int
sel = []
acc = []
wacc = []
for it in range(12):
    idx.search(qd, K, out_scores=s, out_rows=r, stream=st)
    idx.search_finish(st)
    buf = np.zeros(256 * 24, np.uint64)
    assert L.crh_debug_fused_stamps(buf.ctypes.data) == 0
    if it >= 2:
        acc.append(buf[:2048].reshape(256, 8).astype(np.int64))
        wacc.append(buf[2048:].reshape(256, 16).astype(np.int64))
        sb = np.zeros(64 * 8, np.uint64)
        assert L.crh_debug_select_stamps(sb.ctypes.data) == 0
        sel.append(sb.reshape(64, 8).astype(np.int64))
names = ["query image", "sample tile", "wait A", "threshold", "wait B", "main loop"]
a = np.stack(acc)                                    # [runs, wg, stamp]
t0 = a[:, :, 0].min(axis=1, keepdims=True)           # the first workgroup's entry
print("stamp (us after the first workgroup's entry): min / median / max over workgroups, mean over runs")
for i in range(7):
    v = (a[:, :, i] - t0) / 100.0
    print(f"  {i} {(['entry'] + names)[i]:12s} {v.min(axis=1).mean():9.1f} {np.median(v, axis=1).mean():9.1f} {v.max(axis=1).mean():9.1f}")
d = np.diff(a[:, :, :7], axis=2) / 100.0
print("phase durations (us): median over workgroups / max, mean over runs")
for i, n in enumerate(names):
    print(f"  {n:12s} {np.median(d[:, :, i], axis=1).mean():9.1f} {d[:, :, i].max(axis=1).mean():9.1f}")
tauwg = d[:, :B, 3]; rest = d[:, B:, 3]
print(f"threshold phase: workgroups 0..{B-1} (one query each) median {np.median(tauwg):.1f} us, the others {np.median(rest):.1f} us")
w = (np.stack(wacc) - t0[:, :, None]) / 100.0            # [runs, wg, wave]: end of each wave's main loop
print("end of the main loop per wave slot (us after kernel entry), median over workgroups and runs:")
print("  " + " ".join(f"{np.median(w[:, :, i]):7.0f}" for i in range(16)))
print(f"  earliest wave {w.min(axis=(1, 2)).mean():.0f}, median {np.median(w):.0f}, last wave {w.max(axis=(1, 2)).mean():.0f}")
e = (a[:, :, 6] - t0) / 100.0                        # [runs, wg]: wave 0 of each workgroup leaves the main loop
print("end of the main loop by XCD (workgroup % 8): min / median / max, mean over runs")
for x in range(8):
    v = e[:, x::8]
    print(f"  xcd {x}: {v.min(axis=1).mean():7.0f} {np.median(v, axis=1).mean():7.0f} {v.max(axis=1).mean():7.0f}")
qs_ = np.percentile(e, [0, 5, 25, 50, 75, 95, 100], axis=1).mean(axis=1)
print("  all workgroups, percentiles 0/5/25/50/75/95/100: " + " ".join(f"{v:.0f}" for v in qs_))
if (a[:, :, 7] > 0).all():
    v = (a[:, :, 7] - a[:, :, 6]) / 100.0
    print(f"hand-over of the candidates after the main loop (us): median {np.median(v):.1f}, max {v.max(axis=1).mean():.1f}; kernel end {((a[:, :, 7] - t0) / 100.0).max(axis=1).mean():.0f}")
sl = np.stack(sel)                                   # [runs, query, stamp]; 6 = survivors, 7 = candidates
ph = np.diff(sl[:, :, :5], axis=2) / 100.0
print("k_select per query workgroup (us): k-th largest / survivor sweep / canonical re-score / ranking -- median, max over queries; mean over runs")
for i, n in enumerate(["k-th largest", "survivor sweep", "re-score", "ranking"]):
    print(f"  {n:15s} {np.median(ph[:, :, i], axis=1).mean():8.1f} {ph[:, :, i].max(axis=1).mean():8.1f}")
print(f"  candidates per query: median {np.median(sl[:, :, 7]):.0f}, max {sl[:, :, 7].max()};  survivors: median {np.median(sl[:, :, 6]):.0f}, max {sl[:, :, 6].max()}")
