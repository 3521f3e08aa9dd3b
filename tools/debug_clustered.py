#!/usr/bin/env python3
"""Per-query candidate counts of the scan on a clustered corpus (diagnostic)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import coderag_amd
from coderag_amd import ffi
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
from bench_matrix import build_corpus
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
dev = torch.device("cuda", 0)
idx, head, _, centres = build_corpus(torch, ffi, dev, rows, ffi.DTYPE_BF16, "clustered", 0, 1000)
D = 768
qs = np.random.default_rng(7).standard_normal((1024, D)).astype(np.float32)
pick = np.random.default_rng(8).integers(0, 1000, 1024)
q_all = (centres.cpu().numpy()[pick] + 0.7 * qs / np.sqrt(D)).astype(np.float32)
for nq in (64,):
    s, r = idx.search(q_all[:nq], 100)
    print("nq", nq, idx.stats(), flush=True)
cnt = []
for i in range(16):
    t0 = time.perf_counter()
    s, r = idx.search(q_all[i:i + 1], 100)
    st = idx.stats()
    cnt.append(st["max_query_cands"])
    print(i, "cands", st["max_query_cands"], "fallback", st["fallback_used"], "top1 %.4f top100 %.4f" % (s[0, 0], s[0, 99]), "ms %.2f" % ((time.perf_counter() - t0) * 1e3), flush=True)
# Gaussian queries against the clustered corpus
for i in range(4):
    s, r = idx.search(qs[i:i + 1], 100)
    st = idx.stats()
    print("gauss q", i, "cands", st["max_query_cands"], "top1 %.4f top100 %.4f" % (s[0, 0], s[0, 99]), flush=True)
