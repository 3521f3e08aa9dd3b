#!/usr/bin/env python3
"""Where the scans spend their time (k_scan_fused: before the main loop; k_scan_i8: inside the pass and the threshold launch): per-workgroup wall-clock stamps (100 MHz) taken by a measurement
build of the library (tools/build_stamps_lib.sh: -DCRH_FUSED_STAMPS -> lib/libcoderag_hip_stamps.so; not part of build.sh):
0 kernel entry, 1 query image in LDS, 2 sample tile done, 3 past wait A, 4 threshold written, 5 past wait B, 6 main loop done.
CODERAG_HIP_LIB=code-rag_amd/lib/libcoderag_hip_stamps.so python tools/fused_stamps.py [rows]
STAMPS_FILTER=1: the batches carry the filter "code column 0 == 1" over three uniform codes (the bench's `filtered` leg)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import coderag_amd
from coderag_amd import ffi
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda:0"); D, B, K = 768, 64, 100
st = torch.cuda.current_stream().cuda_stream
flt = [(0, 1)] if os.environ.get("STAMPS_FILTER") == "1" else None
idx = ffi.Index(D, ffi.DTYPE_BF16, capacity_rows=rows, device=0, n_code_cols=1 if flt else 0)
gen = torch.Generator(device=dev); gen.manual_seed(20251226)
for r0 in range(0, rows, 500_000):
    m = min(500_000, rows - r0)
    idx.append(torch.randn((m, D), generator=gen, device=dev),
               torch.randint(0, 3, (m, 1), generator=gen, device=dev, dtype=torch.int32) if flt else None, stream=st)
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
    idx.search(qd, K, filters=flt, out_scores=s, out_rows=r, stream=st)
    idx.search_finish(st)
    buf = np.zeros(256 * 24, np.uint64)
    assert L.crh_debug_fused_stamps(buf.ctypes.data) == 0
    if it >= 2:
        acc.append(buf[:2048].reshape(256, 8).astype(np.int64))
        wacc.append(buf[2048:].reshape(256, 16).astype(np.int64))
        sb = np.zeros(64 * 16, np.uint64)
        assert L.crh_debug_select_stamps(sb.ctypes.data) == 0
        sel.append(sb.reshape(64, 16).astype(np.int64))
names = ["query image", "sample tile", "wait A", "threshold", "wait B", "main loop"]
a = np.stack(acc)                                    # [runs, wg, stamp]
t0 = a[:, :, 0].min(axis=1, keepdims=True)           # the first workgroup's entry
three = idx.nomination() == ffi.NOMINATE_INT8        # the int8 scan is three launches: the stamps below are those of its LAST one, the pass
if three:
    print("int8-nominated batches: sample tiles, thresholds and the pass are three launches (their durations: tools/i8_timeline.py on a kernel "
          "trace); the stamps are the pass's -- 0 entry, 5 query image and thresholds in LDS, 6 main loop done -- and the threshold launch's sub-stamps")
    for i, n in ((5, "loop start"), (6, "main loop")):
        v = (a[:, :, i] - t0) / 100.0
        print(f"  {i} {n:12s} {v.min(axis=1).mean():9.1f} {np.median(v, axis=1).mean():9.1f} {v.max(axis=1).mean():9.1f}   (us after the first workgroup's entry: min / median / max)")
    a[:, :, 1:5] = a[:, :, :1]                       # (stamps 1..4 belong to the one-launch scans only)
if not three:
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
wa = np.stack(wacc)
if (wa[:, :B, 8:13] > 0).all():      # k_scan_i8: sub-stamps of the threshold launch's workgroups 0..63 (slots 8..12)
    tt = (wa[:, :B, 9:13] - wa[:, :B, 8:9]) / 100.0
    print("threshold launch, a query's workgroup, us after its keys were in LDS -- k-th largest key / rows selected / their scores / k-th largest score: "
          + " ".join(f"{np.median(tt[:, :, i]):.1f}" for i in range(4)))
    w = w[:, :, :8]
print("end of the main loop per wave slot (us after kernel entry), median over workgroups and runs:")
print("  " + " ".join(f"{np.median(w[:, :, i]):7.0f}" for i in range(w.shape[2])))
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
sl = np.stack(sel)                                   # [runs, query, 16]: stamps 0..7; 8 survivors of a part, 9 candidates, 10 survivors of the query, 11 rows given the canonical chain
i8 = (sl[:, :, 6] > 0).all()
if i8:    # behind the int8 scan: 0 entry, 1 k-th largest lower end, 2 survivor sweep, 3 fast scores, 4 (last part) gathered + cut, 5 canonical chain, 6 ranked
    names_s, last = ["k-th largest", "survivor sweep", "fast scores", "publish + cut", "canonical chain", "ranking"], 7
else:     # bf16 scans: 0 entry, 1 k-th largest, 2 survivor sweep, 3 canonical re-score, 4 ranked
    names_s, last = ["k-th largest", "survivor sweep", "re-score", "ranking"], 5
ph = np.diff(sl[:, :, :last], axis=2) / 100.0
print("k_select per query workgroup (us) -- median, max over queries; mean over runs (stamps 0..3 are those of the part that wrote last)")
for i, n in enumerate(names_s):
    print(f"  {n:15s} {np.median(ph[:, :, i], axis=1).mean():8.1f} {ph[:, :, i].max(axis=1).mean():8.1f}")
print(f"  whole kernel per query: median {np.median((sl[:, :, last - 1] - sl[:, :, 0]) / 100.0):.1f} us")
print(f"  candidates per query: median {np.median(sl[:, :, 9]):.0f}, max {sl[:, :, 9].max()};  survivors of a part: median {np.median(sl[:, :, 8]):.0f}, max {sl[:, :, 8].max()}"
      + (f";  survivors of the query: median {np.median(sl[:, :, 10]):.0f}, max {sl[:, :, 10].max()};  canonical chains: median {np.median(sl[:, :, 11]):.0f}, max {sl[:, :, 11].max()}" if i8 else ""))
