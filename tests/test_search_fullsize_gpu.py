"""BASELINE.json's full size (10M x 768 bf16 resident in HBM, batch-64, top-100) checked through properties that do not
need a 10M-row CPU scan:
  * order      : every result list is sorted by (score desc, row asc), rows unique, all rows alive;
  * planted    : a stored row used as a query comes back first with score ~1;
  * shards     : two 5M-row shards searched separately and merged by crh_merge_topk == the 10M-row search, bit for bit
                 (the 8-GPU configuration in miniature);
  * filter     : a payload filter selecting 1M of the rows == the oracle run on exactly those 1M rows;
  * modes      : nominated from the int8 copy (the default from 1M rows), from the bf16 tiles in one launch and in three: identical bytes;
  * idempotent : the same search twice gives identical bytes;
  * wide       : a 512-query call (two passes of k_scan_wide) == the same queries searched 64 at a time.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, D, K, NQ, BLOCK = 10_000_000, 768, 100, 64, 500_000


def test_full_size_properties(gpu):
    import torch
    import coderag_amd  # noqa: F401
    from coderag_amd import ffi
    from oracle import search as orc
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev)
    gen.manual_seed(20251226)
    full = ffi.Index(D, ffi.DTYPE_BF16, capacity_rows=N, n_code_cols=1)
    halves = [ffi.Index(D, ffi.DTYPE_BF16, capacity_rows=N // 2) for _ in range(2)]
    keep_rows, keep_vecs = [], []
    for r0 in range(0, N, BLOCK):
        xb = torch.randn((BLOCK, D), generator=gen, device=dev, dtype=torch.float32)
        codes = (torch.arange(r0, r0 + BLOCK, device=dev, dtype=torch.int64) % 10).to(torch.int32).reshape(-1, 1).contiguous()
        full.append(xb, codes)
        halves[r0 // (N // 2)].append(xb)
        sel = torch.arange(3 - r0 % 10 if r0 % 10 <= 3 else 13 - r0 % 10, BLOCK, 10, device=dev)     # rows with row % 10 == 3
        keep_rows.append((sel + r0).cpu().numpy())
        keep_vecs.append(xb[sel].cpu().numpy())
        torch.cuda.synchronize()
        del xb
    assert full.count() == (N, N)
    sub_rows = np.concatenate(keep_rows)
    sub = np.concatenate(keep_vecs)
    assert len(sub_rows) == N // 10 and np.all(sub_rows % 10 == 3)

    rng = np.random.default_rng(7)
    q = rng.standard_normal((NQ, D)).astype(np.float32)
    planted = rng.choice(len(sub_rows), 8, replace=False)
    q[:8] = sub[planted]                                            # stored rows as queries
    s, r = full.search(q, K)
    st = full.stats()
    assert st["fallback_used"] == 0 and st["rows"] == N

    # order / uniqueness
    assert np.all(r >= 0) and np.all(r < N)
    assert np.all((s[:, :-1] > s[:, 1:]) | ((s[:, :-1] == s[:, 1:]) & (r[:, :-1] < r[:, 1:])))
    assert all(len(set(row)) == K for row in r)
    # planted
    assert np.array_equal(r[:8, 0], sub_rows[planted]) and np.all(np.abs(s[:8, 0] - 1.0) < 4e-3)
    # the three nomination modes (int8 copy -- the default at this size --, bf16 tiles in one launch, in three) agree bit for bit
    assert full.nomination() == ffi.NOMINATE_INT8
    for mode in (ffi.NOMINATE_BF16, ffi.NOMINATE_BF16_3):
        full.set_nomination(mode)
        assert full.nomination() == mode
        sm, rm = full.search(q, K)
        assert np.array_equal(rm, r) and np.array_equal(sm.view(np.uint32), s.view(np.uint32)), mode
        assert full.stats()["fallback_used"] == 0
    full.set_nomination(ffi.NOMINATE_INT8)
    assert full.nomination() == ffi.NOMINATE_INT8
    # idempotent
    s2, r2 = full.search(q, K)
    assert np.array_equal(r, r2) and np.array_equal(s.view(np.uint32), s2.view(np.uint32))

    # shards + merge == full
    qd = torch.from_numpy(q).to(dev)
    ps = torch.empty((2, NQ, K), dtype=torch.float32, device=dev)
    pr = torch.empty((2, NQ, K), dtype=torch.int64, device=dev)
    for i, h in enumerate(halves):
        h.search(qd, K, row_base=i * (N // 2), out_scores=ps[i], out_rows=pr[i])
        h.search_finish()
    ms = torch.empty((NQ, K), dtype=torch.float32, device=dev)
    mr = torch.empty((NQ, K), dtype=torch.int64, device=dev)
    ffi.merge_topk(ps, pr, ms, mr)
    torch.cuda.synchronize()
    assert np.array_equal(mr.cpu().numpy(), r) and np.array_equal(ms.cpu().numpy().view(np.uint32), s.view(np.uint32))

    # 512 queries in one call = two passes of the wide scan == the same queries 64 at a time, bit for bit
    big = np.concatenate([q, rng.standard_normal((512 - NQ, D)).astype(np.float32)])
    bs, br = full.search(big, K)
    assert full.stats()["batches"] == 2 and full.stats()["fallback_used"] == 0, full.stats()
    assert np.array_equal(br[:NQ], r) and np.array_equal(bs[:NQ].view(np.uint32), s.view(np.uint32))
    for q0 in (64, 256, 448):
        ps_, pr_ = full.search(big[q0:q0 + 64], K)
        assert np.array_equal(br[q0:q0 + 64], pr_) and np.array_equal(bs[q0:q0 + 64].view(np.uint32), ps_.view(np.uint32))
    assert np.all((bs[:, :-1] > bs[:, 1:]) | ((bs[:, :-1] == bs[:, 1:]) & (br[:, :-1] < br[:, 1:])))

    # filter == oracle on the selected 1M rows (8 queries keep the CPU side to a few seconds)
    fs, fr = full.search(q[8:16], K, filters=[(0, 3)])
    es, er = orc.cosine_search(sub, q[8:16], K, bf16=True)
    assert np.array_equal(fr, sub_rows[er]) and np.array_equal(fs.view(np.uint32), es.view(np.uint32))
    # tombstoning the best hit of a query promotes the next one
    full.tombstone(r[20, :1])
    s3, r3 = full.search(q[20:21], K)
    assert np.array_equal(r3[0, : K - 1], r[20, 1:]) and r3[0, K - 1] not in r[20]
    for h in [full] + halves:
        h.close()


def _kind_corpus(torch, kind, n, seed):
    """Rows and queries of one corpus kind: bench.corpus_generator's, plus `shared-mean` -- one common direction with ISOTROPIC
    noise (norm of the mean unit vector 0.86 like encoder output, but scores packed three times closer than its low-rank noise
    packs them): the widest candidate sets the int8 intervals meet, where the one regrowth of its buffers has to happen."""
    import bench
    dev = torch.device("cuda:0")
    if kind != "shared-mean":
        return bench.corpus_generator(torch, dev, kind, seed, D)
    r0 = np.random.default_rng(seed)
    c = r0.standard_normal(D)
    c /= np.linalg.norm(c)
    c_d = torch.from_numpy(c.astype(np.float32)).to(dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    sig = (0.352 / D) ** 0.5
    return (lambda m: c_d[None, :] + sig * torch.randn((m, D), generator=gen, device=dev),
            lambda nq, qseed: (c[None, :] + sig * np.random.default_rng(qseed).standard_normal((nq, D))).astype(np.float32))


@pytest.mark.parametrize("kind", ["clustered", "anisotropic", "shared-mean"])
def test_corpus_kinds_at_2m_rows_against_the_oracle(gpu, kind):
    """What the int8 intervals nominate depends on the data (SURVEY 8(d)'s clustered corpus; encoder-like rows with one shared
    direction).  2M rows per kind, nominated from the int8 copy, ids and f32 score bits against the oracle; the candidate
    buffers may grow once (bit 0 of fallback_used), the copy must not give way to the bf16 scan (bit 2) nor a wait time out."""
    import torch
    import coderag_amd  # noqa: F401
    from coderag_amd import ffi
    from oracle import search as orc
    n, nq = 2_000_000, 16
    block_of, queries_of = _kind_corpus(torch, kind, n, 4242)
    idx = ffi.Index(D, ffi.DTYPE_BF16, capacity_rows=n)
    parts = []
    for r0 in range(0, n, BLOCK):
        xb = block_of(BLOCK)
        idx.append(xb)
        parts.append(xb.cpu().numpy())
        torch.cuda.synchronize()
        del xb
    x = np.concatenate(parts)
    del parts
    q = queries_of(nq, 99)
    assert idx.nomination() == ffi.NOMINATE_INT8
    s, r = idx.search(q, K)
    st = idx.stats()
    print(kind, st)
    assert st["fallback_used"] & 6 == 0, st
    es, er = orc.cosine_search(x, q, K, bf16=True)
    assert np.array_equal(r, er), "ids differ from the oracle"
    assert np.array_equal(s.view(np.uint32), es.view(np.uint32)), "score bits differ from the oracle"
    s2, r2 = idx.search(q, K)                       # buffers are settled now: no regrowth the second time
    assert idx.stats()["fallback_used"] == 0 and np.array_equal(r2, r) and np.array_equal(s2.view(np.uint32), s.view(np.uint32))
    idx.close()


def test_search_beside_an_encoder_forward_on_another_stream(gpu):
    """The product runs the encoder (provider thread) and the store (its own thread) in one process
    (/root/reference/src/lattice/providers/unixcoder_provider.py:260: the 1-worker executor beside the query path).  The
    one-launch scans need their whole grid resident at two grid-wide waits; beside another stream's kernels a workgroup may not
    be.  Here searches go out on stream A while a 65 k-token packed encoder forward runs on stream B: every result is bit-exact
    (whether the waits held or a batch was recovered in the three-launch form), and a search costs at most 10 ms more than alone."""
    import time
    import torch
    import coderag_amd  # noqa: F401
    from coderag_amd import encoder as drv, ffi
    from oracle import search as orc
    dev = torch.device("cuda:0")
    n, nq, k = 1_000_000, 64, 100
    gen = torch.Generator(device=dev)
    gen.manual_seed(77)
    idx = ffi.Index(D, ffi.DTYPE_BF16, capacity_rows=n)
    parts = []
    for r0 in range(0, n, BLOCK):
        xb = torch.randn((BLOCK, D), generator=gen, device=dev)
        idx.append(xb)
        parts.append(xb.cpu().numpy())
        del xb
    x = np.concatenate(parts)
    q = np.random.default_rng(78).standard_normal((nq, D)).astype(np.float32)
    es, er = orc.cosine_search(x, q[:8], k, bf16=True)
    qd = torch.from_numpy(q).to(dev)
    cfg = drv.EncoderConfig()
    model = drv.HipUniXcoder(drv.synthetic_weights(cfg, 23), cfg, drv.HashTokenizer(cfg.vocab_size), 0)
    rng = np.random.default_rng(79)
    lengths = np.full(256, 256, np.int64)                                       # 65 536 tokens: ~13 ms of back-to-back kernels
    rows = [np.concatenate([[0, 5, 2], rng.integers(16, cfg.vocab_size, int(L) - 4), [2]]).astype(np.int32) for L in lengths]
    flat, off, Lmax = model.pack_rows(rows, list(range(len(rows))))
    ids_d, off_d = torch.from_numpy(flat).to(dev), torch.from_numpy(off).to(dev)
    # the searcher asks for a HIGH-PRIORITY stream: the device then takes its kernels ahead of the forward's ~1000 queued launches at
    # the next kernel boundary (on equal priority a search was seen to sit behind the whole forward, 13 ms, without any wait timing out)
    sA, sB = torch.cuda.Stream(dev, priority=-1), torch.cuda.Stream(dev)
    out_s = [torch.empty((nq, k), dtype=torch.float32, device=dev) for _ in range(8)]
    out_r = [torch.empty((nq, k), dtype=torch.int64, device=dev) for _ in range(8)]
    with torch.cuda.stream(sB):
        ref_emb = model.forward_packed(ids_d, off_d, Lmax).clone()             # warm-up + the embedding every later forward must repeat
    torch.cuda.synchronize()

    def one_search(i):
        t0 = time.perf_counter()
        idx.search(qd, k, out_scores=out_s[i], out_rows=out_r[i], stream=sA.cuda_stream)
        idx.search_finish(sA.cuda_stream)
        return (time.perf_counter() - t0) * 1e3
    alone = [one_search(i) for i in range(8)][2:]
    assert idx.stats()["fallback_used"] == 0 and idx.nomination() == ffi.NOMINATE_INT8
    beside, fallbacks = [], 0
    for rep in range(3):
        with torch.cuda.stream(sB):
            emb = model.forward_packed(ids_d, off_d, Lmax)                      # ~1000 launches queued on B; returns while they run
        for i in range(8):
            beside.append(one_search(i))
            fallbacks |= idx.stats()["fallback_used"]
            got_s, got_r = out_s[i][:8].cpu().numpy(), out_r[i][:8].cpu().numpy()
            assert np.array_equal(got_r, er) and np.array_equal(got_s.view(np.uint32), es.view(np.uint32)), (rep, i)
        sB.synchronize()
        assert torch.equal(emb, ref_emb)                                        # the encoder is not disturbed either
    print(f"search alone {np.median(alone):.3f} ms (max {max(alone):.3f}); beside the forward median {np.median(beside):.3f} ms, max {max(beside):.3f} ms; "
          f"fallback bits seen {fallbacks}")
    assert fallbacks & ~2 == 0                                                  # bit 1 (a wait timed out, batch recovered) may or may not appear
    # a search must not SYSTEMATICALLY wait out a forward (13 ms; on an equal-priority stream it did): the median stays within a few ms of
    # a search alone, and at most one of the 24 may hit a hiccup of the (shared) box -- one 31.9 ms outlier was seen once in round 5 among
    # otherwise 0.8-1.6 ms, on code that passed before and after
    assert np.median(beside) <= max(alone) + 5.0, (np.median(beside), max(alone))
    assert sum(1 for t in beside if t > max(alone) + 10.0) <= 1, sorted(beside)[-3:]
    idx.close()
