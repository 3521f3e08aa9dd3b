#!/usr/bin/env python3
"""Experiment: variants of k_scan_wide (CODERAG_HIP_WIDE_VARIANT) at nq = 65 and 256.  Variants 3/4 compute nothing (results invalid)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("WIDE_CHILD") == "1":
    sys.path.insert(0, ROOT)
    import numpy as np, torch, time
    import coderag_amd
    from coderag_amd import ffi
    rows, D, K = 10_000_000, 768, 100
    dev = torch.device("cuda:0")
    idx = ffi.Index(D, ffi.DTYPE_BF16, capacity_rows=rows)
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    for r0 in range(0, rows, 500_000):
        idx.append(torch.randn((500_000, D), generator=gen, device=dev)); torch.cuda.synchronize()
    qs = torch.from_numpy(np.random.default_rng(7).standard_normal((256, D)).astype(np.float32)).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    for nq in (65, 128, 256):
        s = torch.empty((nq, K), dtype=torch.float32, device=dev); r = torch.empty((nq, K), dtype=torch.int64, device=dev)
        for _ in range(3):
            idx.search(qs[:nq], K, out_scores=s, out_rows=r, stream=st)
        try:
            idx.search_finish(st)
        except Exception as e:
            print("finish:", e, file=sys.stderr)
        torch.cuda.synchronize(); idx.set_profiling(True)
        for _ in range(8):
            idx.search(qs[:nq], K, out_scores=s, out_rows=r, stream=st)
        try:
            idx.search_finish(st)
        except Exception as e:
            print("finish:", e, file=sys.stderr)
        torch.cuda.synchronize()
        ms, n = idx.profile(); idx.set_profiling(False)
        print(json.dumps({"variant": os.environ.get("CODERAG_HIP_WIDE_VARIANT", "0"), "nq": nq, "scan_ms": ms / max(1, n), "GBps": rows * D * 2 / (ms / max(1, n) * 1e-3) / 1e9}), flush=True)
else:
    for v in (sys.argv[1:] or ("0", "1", "2", "3", "4")):
        env = dict(os.environ, CODERAG_HIP_WIDE_VARIANT=v, WIDE_CHILD="1")
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=env)
