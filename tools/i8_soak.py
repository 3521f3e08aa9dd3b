#!/usr/bin/env python3
"""Soak of the int8 nomination against the three-launch bf16 scan on the SAME index (no oracle: the two forms share only the
canonical selection, so a row the intervals wrongly excluded shows as a difference): many batches of varied queries -- Gaussian,
sparse, stored rows, scaled sums of rows, near-duplicates, saturated sign patterns against saturated rows --, varied k, filters and tombstones, three widths, both stores.
python tools/i8_soak.py [batches per configuration]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CODERAG_HIP_I8_MIN_ROWS"] = "0"
import numpy as np
import coderag_amd
from coderag_amd import ffi
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(2026)
bad = total = gave_way = 0
for dim, n in ((768, 300_000), (384, 200_000), (1024, 150_000), (1536, 120_000)):
    for dtype in (ffi.DTYPE_BF16, ffi.DTYPE_F32):
        x = rng.standard_normal((n, dim), dtype=np.float32)
        # a tenth of the rows: heavy-tailed elements, a few dominated by one element, some exact duplicates, some tiny
        hv = rng.choice(n, n // 10, replace=False)
        x[hv] *= np.exp(rng.standard_normal((len(hv), dim))).astype(np.float32)
        sp = rng.choice(n, n // 100, replace=False)
        x[sp, rng.integers(0, dim, len(sp))] += 60.0
        x[rng.choice(n, 2000, replace=False)] = x[rng.choice(n, 2000, replace=False)]
        x[rng.choice(n, 500, replace=False)] *= 1e-12
        # saturated rows: every element at +-c (their int8 images are all +-127: the integer dots reach 127 * 127 * dim, beyond
        # 2^24 at dim 1536 -- the rounding allowance of the intervals, crh_i8.hpp kDotRound), a few signs flipped per row
        pool = rng.choice([-1.0, 1.0], (64, dim)).astype(np.float32)
        sat = rng.choice(n, n // 20, replace=False)
        rows_ = pool[rng.integers(0, 64, len(sat))] * rng.choice([1.0, -1.0, 0.2], (len(sat), 1)).astype(np.float32)
        for i in range(len(sat)):
            f = int(rng.integers(0, 30))
            if f:
                rows_[i, rng.choice(dim, f, replace=False)] *= -1.0
        x[sat] = rows_
        codes = rng.integers(0, 4, (n, 1)).astype(np.int32)
        idx = ffi.Index(dim, dtype, capacity_rows=n, n_code_cols=1)
        idx.append(x, codes)
        idx.tombstone(rng.choice(n, n // 9, replace=False))
        nqmax = 32 if dim == 1536 else 64
        for b in range(nb):
            nq = int(rng.integers(1, nqmax + 1))
            kind = b % 6
            if kind == 5:        # saturated queries: a sign pattern of the pool (the saturated rows' near neighbours), constant magnitude
                q = pool[rng.integers(0, 64, nq)] * np.float32(0.5)
            elif kind == 0:
                q = rng.standard_normal((nq, dim), dtype=np.float32)
            elif kind == 1:
                q = np.zeros((nq, dim), np.float32)
                q[np.arange(nq)[:, None], rng.integers(0, dim, (nq, 6))] = rng.standard_normal((nq, 6)).astype(np.float32)
            elif kind == 2:
                q = x[rng.integers(0, n, nq)] * rng.uniform(0.01, 100.0, (nq, 1)).astype(np.float32)
            elif kind == 3:
                q = (x[rng.integers(0, n, nq)] + x[rng.integers(0, n, nq)] - 0.5 * x[rng.integers(0, n, nq)]).astype(np.float32)
            else:
                q = x[rng.integers(0, n, nq)] + 1e-3 * rng.standard_normal((nq, dim), dtype=np.float32)
            k = int(rng.choice([1, 5, 10, 50, 100, 200, 256]))
            flt = [(0, int(rng.integers(0, 4)))] if b % 3 == 0 else None
            idx.set_nomination(ffi.NOMINATE_INT8)
            s8, r8 = idx.search(q, k, filters=flt)
            st8 = idx.stats()["fallback_used"]
            gave_way += 1 if st8 & 4 else 0
            idx.set_nomination(ffi.NOMINATE_BF16_3)
            s3, r3 = idx.search(q, k, filters=flt)
            total += 1
            if not (np.array_equal(r8, r3) and np.array_equal(s8.view(np.uint32), s3.view(np.uint32))):
                bad += 1
                d = np.argwhere(r8 != r3)
                print(f"MISMATCH dim {dim} dtype {dtype} batch {b} kind {kind} nq {nq} k {k} filter {flt}: first at {d[:3].tolist()} fallback {st8}", flush=True)
        print(f"dim {dim} dtype {'bf16' if dtype == ffi.DTYPE_BF16 else 'f32'}: {nb} batches compared", flush=True)
        idx.close()
print(f"{total} batches, {bad} mismatches; {gave_way} of them overflowed the int8 candidate buffers and were answered by the bf16 scan")
sys.exit(1 if bad else 0)
