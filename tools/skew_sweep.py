#!/usr/bin/env python3
"""Start offsets between co-resident workgroups: does de-phasing pay?  (debug library: crh_debug_set_skew)
  attention: the workgroups sharing a CU start `a` x 1024 cycles apart (k_attn);   GEMM: every other workgroup of an XCD label
  starts `g` x 1024 cycles late (k_gemm_pp).  Interleaved rounds in ONE process, random data, ~65k tokens per launch.
python tools/skew_sweep.py [attn|gemm|both]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import coderag_amd
from coderag_amd import ffi
what = sys.argv[1] if len(sys.argv) > 1 else "both"
dev = torch.device("cuda:0"); Lb = ffi.debug_lib(); H = 12


def timed(call, reps=20):
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


if what in ("attn", "both"):
    skews = (0, 1, 2, 3, 4, 6, 8)
    for L in (72, 128, 160, 208, 256, 320, 384, 512):
        B = max(1, 65536 // L); T = B * L
        qkv = torch.randn((T, 3 * H * 64), device=dev).to(torch.bfloat16)
        out = torch.empty((T, H * 64), dtype=torch.bfloat16, device=dev)
        Lmax = (L + 15) // 16 * 16; nw = (Lmax + 63) // 64
        km = torch.zeros((B, nw), dtype=torch.int64)
        for w in range(nw):
            bits = min(64, max(0, L - 64 * w))
            km[:, w] = -1 if bits == 64 else (1 << bits) - 1
        km = km.to(dev); off = torch.arange(0, T + 1, L, dtype=torch.int32, device=dev)
        res = {s: [] for s in skews}
        ref = None
        for rnd in range(4):
            for s in skews:
                ffi.check(Lb.crh_debug_set_skew(s, 0), Lb)
                res[s].append(timed(lambda: ffi.check(Lb.crh_attn_fwd_packed(qkv.data_ptr(), off.data_ptr(), km.data_ptr(), out.data_ptr(), B, T, Lmax, H, 0), Lb)))
                if ref is None:
                    ref = out.clone()
                assert torch.equal(ref, out)
        print(f"attn L={L:3d}: " + "  ".join(f"skew {s}: {np.median(res[s]):6.1f}" for s in skews) + " us", flush=True)
    ffi.check(Lb.crh_debug_set_skew(0, 0), Lb)

if what in ("gemm", "both"):
    T = 65536
    g = torch.Generator(device="cpu").manual_seed(1)
    skews = (0, 1, 2, 4, 6, 10, 16)
    for name, N, K, act in (("qkv", 2304, 768, 0), ("oproj", 768, 768, 0), ("ffn1", 3072, 768, 1), ("ffn2+res", 768, 3072, 2)):
        a = torch.randn((T, K), generator=g).to(dev, torch.bfloat16)
        w = (torch.randn((N, K), generator=g) / K ** 0.5).to(dev, torch.bfloat16)
        b = torch.randn((N,), generator=g).to(dev)
        y = torch.empty((T, N), dtype=torch.bfloat16, device=dev)
        res = {s: [] for s in skews}
        for rnd in range(4):
            for s in skews:
                ffi.check(Lb.crh_debug_set_skew(0, s), Lb)
                if act == 2:
                    call = lambda: ffi.check(Lb.crh_debug_gemm_variant(a.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), T, N, K, 19, 0), Lb)
                else:
                    call = lambda: ffi.check(Lb.crh_gemm_bf16_bias(a.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), T, N, K, act, 0), Lb)
                res[s].append(timed(call))
        fl = 2.0 * T * N * K
        print(f"gemm {name:8s}: " + "  ".join(f"skew {s}: {np.median(res[s]):6.1f}" for s in skews) + f" us   (best {fl / min(np.median(v) for v in res.values()) / 1e6:.0f} TFLOP/s)", flush=True)
    ffi.check(Lb.crh_debug_set_skew(0, 0), Lb)
