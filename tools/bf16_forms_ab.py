#!/usr/bin/env python3
"""One launch with two grid-wide waits (NOMINATE_BF16, `k_scan_fused`) against three launches (NOMINATE_BF16_3: seed scan,
thresholds, pass) for the bf16-tile nomination -- the library's default below 1M rows.  Round-4 review item 2: the one-launch
form stays the default only where it is >= 3 % faster.

Per corpus size and store dtype, on one index:
  alone    ms per 64-query top-100 batch, 40 batches after 10, wall clock around a synchronize; the two forms alternate
           A B A B ... over 6 rounds (a drifting clock hits both alike); median of the rounds
  beside   single searches (launch + finish, wall clock) on a high-priority stream while a 65 k-token packed encoder forward
           runs on another stream (the reference indexes while it serves: pipeline/watcher.py:260-263); median / max per form,
           and the fallback bits the index reported (bit 1 = a grid-wide wait timed out and the batch was re-run)

    python tools/bf16_forms_ab.py [rows ...]        # default 30000 100000 300000 1000000
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CODERAG_HIP_I8"] = "0"           # no int8 copy: this is about the bf16-tile forms
import numpy as np
import torch
import coderag_amd  # noqa: F401
from coderag_amd import encoder as drv, ffi

sizes = [int(a) for a in sys.argv[1:]] or [30_000, 100_000, 300_000, 1_000_000]
dev = torch.device("cuda:0")
D, B, K = 768, 64, 100
gen = torch.Generator(device=dev)
gen.manual_seed(11)
FORMS = ((ffi.NOMINATE_BF16, "one launch"), (ffi.NOMINATE_BF16_3, "three launches"))

cfg = drv.EncoderConfig()
model = drv.HipUniXcoder(drv.synthetic_weights(cfg, 23), cfg, drv.HashTokenizer(cfg.vocab_size), 0)
rng = np.random.default_rng(79)
rows_tok = [np.concatenate([[0, 5, 2], rng.integers(16, cfg.vocab_size, 252), [2]]).astype(np.int32) for _ in range(256)]
flat, off, Lmax = model.pack_rows(rows_tok, list(range(len(rows_tok))))
ids_d, off_d = torch.from_numpy(flat).to(dev), torch.from_numpy(off).to(dev)
sA, sB = torch.cuda.Stream(dev, priority=-1), torch.cuda.Stream(dev)
with torch.cuda.stream(sB):
    model.forward_packed(ids_d, off_d, Lmax)
torch.cuda.synchronize()


def batches(idx, qd, s, r, stream, steps=40, warm=10):
    for _ in range(warm):
        idx.search(qd, K, out_scores=s, out_rows=r, stream=stream)
    idx.search_finish(stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        idx.search(qd, K, out_scores=s, out_rows=r, stream=stream)
    idx.search_finish(stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def one(idx, qd, s, r, stream):
    t0 = time.perf_counter()
    idx.search(qd, K, out_scores=s, out_rows=r, stream=stream)
    idx.search_finish(stream)
    return (time.perf_counter() - t0) * 1e3


print(f"{'rows':>9s} {'store':>5s} | alone ms/batch: one launch, three launches, one/three | single search alone: one, three | "
      f"beside a 65k-token forward: one med/max (fallback bits), three med/max (fallback bits)")
for n in sizes:
    for dtype, dname in ((ffi.DTYPE_BF16, "bf16"), (ffi.DTYPE_F32, "f32")):
        idx = ffi.Index(D, dtype, capacity_rows=n, device=0)
        st0 = torch.cuda.current_stream().cuda_stream
        for r0 in range(0, n, 250_000):
            idx.append(torch.randn((min(250_000, n - r0), D), generator=gen, device=dev), stream=st0)
        torch.cuda.synchronize()
        qd = torch.randn((B, D), generator=gen, device=dev)
        s = torch.empty((B, K), dtype=torch.float32, device=dev)
        r = torch.empty((B, K), dtype=torch.int64, device=dev)
        ref = None
        alone = {m: [] for m, _ in FORMS}
        for rnd in range(6):
            for m, _ in FORMS:
                idx.set_nomination(m)
                alone[m].append(batches(idx, qd, s, r, st0))
                assert idx.nomination() == m, (idx.nomination(), m)
                got = (s.clone(), r.clone())
                if ref is None:
                    ref = got
                assert torch.equal(got[1], ref[1]) and torch.equal(got[0].view(torch.int32), ref[0].view(torch.int32)), "the two forms disagree"
        single, beside, bits = {}, {}, {}
        for m, _ in FORMS:
            idx.set_nomination(m)
            single[m] = [one(idx, qd, s, r, sA.cuda_stream) for _ in range(12)][4:]
        for m, _ in FORMS:
            beside[m], bits[m] = [], 0
        for rep in range(4):
            for m, _ in FORMS:
                idx.set_nomination(m)
                with torch.cuda.stream(sB):
                    model.forward_packed(ids_d, off_d, Lmax)
                for _ in range(8):
                    beside[m].append(one(idx, qd, s, r, sA.cuda_stream))
                    bits[m] |= idx.stats()["fallback_used"]
                    assert torch.equal(r, ref[1]) and torch.equal(s.view(torch.int32), ref[0].view(torch.int32)), "a search beside the forward differs"
                sB.synchronize()
        a1, a3 = np.median(alone[FORMS[0][0]]), np.median(alone[FORMS[1][0]])
        print(f"{n:>9d} {dname:>5s} | {a1:.4f} {a3:.4f} {a1 / a3:.3f} | {np.median(single[1]):.3f} {np.median(single[0]):.3f} | "
              f"{np.median(beside[1]):.3f}/{max(beside[1]):.3f} ({bits[1]})  {np.median(beside[0]):.3f}/{max(beside[0]):.3f} ({bits[0]})", flush=True)
        idx.close()
        torch.cuda.empty_cache()
