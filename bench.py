#!/usr/bin/env python3
"""bench.py -- headline benchmark: cosine top-100 queries/s over a 10M x 768 bf16 corpus resident in HBM.

A "step" is one 64-query batch searched against this rank's corpus shard (BASELINE.json configs[2]:
"1xMI355X: 10Mx768 bf16 corpus resident in HBM, batch-64 queries").  With N ranks every rank holds its
own 10M-row shard (weak scaling, configs[3]); a step additionally all-gathers the N local top-100
lists over RCCL and merges them on every rank.  `value` counts (query, 10M-row shard) scans per second
over all ranks, i.e. row.query pairs/s / 1e7.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  No reference code or CPU path is inside the timed region; the CPU
baseline leg (oracle, test infrastructure) runs afterwards on rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

_T0 = time.perf_counter()


def log(msg: str) -> None:
    """Progress on stderr (the JSON line on stdout stays alone); also keeps long runs visibly alive."""
    print(f"[bench +{time.perf_counter() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def host_threads() -> int:
    """Threads for the CPU baseline legs: the box's CPU share for one GPU is 16 cores even when more are visible."""
    return max(1, min(16, os.cpu_count() or 1))


HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def main() -> None:
    # stdout carries exactly ONE line, the JSON: libraries that print there (RCCL writes a version banner at communicator
    # creation) are sent to stderr for the whole run, and the line is written to the saved descriptor at the end
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rows", type=int, default=10_000_000, help="corpus rows per GPU")
    ap.add_argument("--queries", type=int, default=64)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--dtype", choices=["bf16", "f32"], default="bf16")
    ap.add_argument("--seed-tiles", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--embed-chunks", type=int, default=100000, help="synthetic chunks for the encoder leg (0 = skip; BASELINE configs[1]: 100k synthetic code chunks)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --rows per GPU (BASELINE C4a); strong: --rows in total, split over the ranks (C4b)")
    ap.add_argument("--check-rows", type=int, default=200_000, help="rows of the parity subsample checked vs the oracle")
    args = ap.parse_args()

    import numpy as np
    import torch
    import coderag_amd  # noqa: F401
    from coderag_amd import ffi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # CODERAG_BENCH_FORCE_DIST=1 runs the N>1 code path (RCCL init, all-gather, merge, max-over-ranks) with one rank -- the
    # rehearsal available on a one-GPU box
    force_dist = os.environ.get("CODERAG_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force_dist and world == 1:
            os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    D, N, B, K = 768, args.rows, args.queries, args.k
    if args.scaling == "strong":
        N = (args.rows + world - 1) // world
    dtype = ffi.DTYPE_BF16 if args.dtype == "bf16" else ffi.DTYPE_F32
    elem = 2  # the scan always streams the bf16 tiled copy
    stream = torch.cuda.current_stream().cuda_stream

    # ---- synthetic corpus, generated on the device in blocks (never staged through host lists)
    idx = ffi.Index(D, dtype, capacity_rows=N, device=local_rank)
    if args.seed_tiles:
        idx.set_tuning(seed_tiles=args.seed_tiles)
    gen = torch.Generator(device=dev)
    gen.manual_seed(20251226 + rank)
    block = 500_000
    head = None
    for r0 in range(0, N, block):
        m = min(block, N - r0)
        xb = torch.randn((m, D), generator=gen, device=dev, dtype=torch.float32)
        idx.append(xb, stream=stream)
        if r0 == 0:
            head = xb[: min(args.check_rows, m)].cpu().numpy()
        torch.cuda.synchronize()
        del xb
    log(f"corpus resident: {N} rows x {D} ({args.dtype})")
    qs = np.random.default_rng(7).standard_normal((B, D)).astype(np.float32)
    qd = torch.from_numpy(qs).to(dev)
    row_base = rank * N

    # per-rank result records [scores | rows] (ffi.topk_exchange_buffers): the exchange is ONE all-gather per step
    nslots = 4
    slots = [ffi.topk_exchange_buffers(torch, world, B, K, dev) for _ in range(nslots)]
    out_s, out_r = [sl[1] for sl in slots], [sl[2] for sl in slots]
    multi = dist is not None
    if multi:
        mer_s = torch.empty((B, K), dtype=torch.float32, device=dev)
        mer_r = torch.empty((B, K), dtype=torch.int64, device=dev)

    # per-step device time stamps on the launch stream (torch's current stream): step i spans ev[i] .. ev[i+1]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    ev_x = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)] if dist is not None else None

    def step(i: int, timed: bool = False) -> None:
        s, r = out_s[i % nslots], out_r[i % nslots]
        idx.search(qd, K, row_base=row_base, out_scores=s, out_rows=r, stream=stream)
        if dist is not None:
            if timed:
                ev_x[i].record()
            local, _, _, gathered, gat_s, gat_r = slots[i % nslots]
            dist.all_gather_into_tensor(gathered.view(-1), local)
            ffi.merge_topk(gat_s, gat_r, mer_s, mer_r, stream)
        if timed:
            ev[i + 1].record()

    def fence() -> None:
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    idx.search_finish(stream)
    fence()
    idx.set_profiling(True)
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        step(i, True)
    idx.search_finish(stream)  # also verifies no candidate buffer overflowed in any timed step
    fence()
    dt = time.perf_counter() - t0
    log(f"search timed: {args.steps} steps in {dt:.3f} s")
    per_step = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)])
    exchange = np.array([ev_x[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]) if dist is not None else None
    scan_ms_total, scan_launches = idx.profile()
    idx.set_profiling(False)
    # the same tile walk with its loads alone (no MFMA, no candidates): what this access pattern can read on this device
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        ffi.check(ffi.lib().crh_debug_read_ceiling(idx._handle(), stream))
    e0.record()
    for _ in range(10):
        ffi.check(ffi.lib().crh_debug_read_ceiling(idx._handle(), stream))
    e1.record()
    torch.cuda.synchronize()
    probe_ms = e0.elapsed_time(e1) / 10
    stats = idx.stats()

    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- parity on a subsample (outside the timed region): first check_rows rows of rank 0 vs the oracle
    parity = None
    if rank == 0 and head is not None and args.check_rows > 0:
        from oracle import search as orc
        sub = ffi.Index(D, dtype, capacity_rows=head.shape[0], device=local_rank)
        sub.append(head)
        gs, gr = sub.search(qs, K)
        es, er = orc.cosine_search(head, qs, K, bf16=(args.dtype == "bf16"))
        truth_s, truth_r = orc.cosine_search(head, qs, K, bf16=False)
        recall_same = float(np.mean([len(set(a) & set(b)) / K for a, b in zip(gr, er)]))
        recall_f32 = float(np.mean([len(set(a) & set(b)) / K for a, b in zip(gr, truth_r)]))
        parity = {"rows": int(head.shape[0]), "ids_bit_exact": bool(np.array_equal(gr, er)),
                  "scores_bit_exact": bool(np.array_equal(gs.view(np.uint32), es.view(np.uint32))),
                  "recall_at_k_vs_oracle_same_precision": recall_same, "recall_at_k_vs_f32_truth": recall_f32}
        sub.close()

    # HBM traffic of the scan kernel comes from PMC counters, which need their own rocprofv3 passes (tools/gpu_prof.sh);
    # the corrected per-launch figure of the committed pass is reported when it was taken at this corpus size
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_scan.json")))
        if int(pmc.get("rows", -1)) == N:
            traffic = float(pmc["hbm_read_bytes_corrected"]) + float(pmc["hbm_write_bytes"])
    except (OSError, ValueError, KeyError):
        pass
    ms_per_step = dt * 1e3 / args.steps
    value = world * B * args.steps / dt * (N / 1e7)
    scan_ms = scan_ms_total / max(1, scan_launches)
    alg_bytes = float(N) * D * elem
    achieved = alg_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    out = {
        "metric": "top-k queries/s over 10Mx768 (cosine top-100, batch 64)",
        "value": value,
        "unit": "queries/s per 10M-row shard (row.query pairs/s / 1e7)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "bf16" if args.dtype == "bf16" else "f32",
        "data": "synthetic: N(0,I) rows generated on device (torch seed 20251226+rank), normalised on insert; queries numpy default_rng(7)",
        "config": {"workload": f"{world}x MI355X: {N}x{D} {args.dtype} corpus per GPU resident in HBM, batch-{B} queries, exact top-{K}"
                               + (", all-gather + merge of per-GPU top-k over RCCL" if world > 1 else "")
                               + (f" ({args.rows} rows in total, strong scaling)" if args.scaling == "strong" else ""),
                   "rows_per_gpu": N, "dim": D, "batch": B, "k": K, "parallelism": f"row-shard x{world}",
                   "precision": ("bf16 corpus and queries on MFMA, f32 accumulate, canonical f32 re-score of the survivors" if args.dtype == "bf16"
                                 else "f32 store: bf16 MFMA scan nominates, f32 canonical re-score decides")},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": "profiles/pmc_scan.json (separate rocprofv3 --pmc passes; FETCH_SIZE x2 per the gfx950 guide)" if traffic else None,
                     "loads_only_probe": {"GBps": alg_bytes / (probe_ms * 1e-3) / 1e9, "ms": probe_ms,
                                          "frac_of_it": (probe_ms / scan_ms) if scan_ms > 0 else None,
                                          "what": "crh_debug_read_ceiling: the scan's grid and nt loads without MFMA or candidate logic"},
                     "kernel": "k_scan<48,1,16,8>", "kernel_ms": scan_ms, "launches": scan_launches,
                     "algorithmic_bytes_per_launch": alg_bytes},
        "step_ms_device": {"median": float(np.median(per_step)), "p10": float(np.percentile(per_step, 10)),
                           "p90": float(np.percentile(per_step, 90))},
        "exchange_ms_device": ({"median": float(np.median(exchange)), "p10": float(np.percentile(exchange, 10)),
                                "p90": float(np.percentile(exchange, 90)),
                                "what": "1 RCCL all-gather of the [scores | rows] records + k_merge_topk, rank 0"} if exchange is not None else None),
        "search_stats": stats,
        "parity": parity,
    }

    log("parity subsample checked" if parity else "parity subsample skipped")
    if args.embed_chunks > 0:
        idx.close()
        emb = embed_leg(np, torch, local_rank, args.embed_chunks, rank, world, dist, cpu=(rank == 0 and world == 1 and not args.no_cpu_baseline))
        if rank == 0:
            out["embed"] = emb
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(np, B, K, D)
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    idx.close()
    if dist is not None:
        dist.destroy_process_group()


MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16


def embed_leg(np, torch, local_rank, n_chunks, rank, world, dist, cpu):
    """Second half of BASELINE.json's metric: chunks embedded/s by the HIP UniXcoder encoder (configs[1] shape:
    synthetic chunks, lengths ~ clip(round(exp(N(ln 160, 0.8^2))), 8, 512), seeded random RoBERTa-base weights in bf16).
    Each rank embeds its own n_chunks (weak scaling, no collective).  FLOPs are counted on TRUE lengths."""
    from coderag_amd import encoder as drv
    cfg = drv.EncoderConfig()
    model = drv.HipUniXcoder(drv.synthetic_weights(cfg, 23), cfg, drv.HashTokenizer(cfg.vocab_size), local_rank)
    rng = np.random.default_rng(1234 + rank)
    lengths = np.clip(np.round(np.exp(rng.normal(np.log(160.0), 0.8, n_chunks))), 8, 512).astype(np.int64)
    dev = torch.device("cuda", local_rank)
    batches = []
    for rows, L in model.plan_batches(lengths, max_tokens=65536):
        host = np.full((len(rows), L), cfg.pad_token_id, dtype=np.int32)
        for r, i in enumerate(rows):
            n = int(lengths[i])
            host[r, :n] = np.concatenate([[0, 5, 2], rng.integers(16, cfg.vocab_size, n - 4), [2]]) if n >= 4 else [0, 5, 2, 2][:n]
        batches.append(torch.from_numpy(host).to(dev))
    log(f"encoder leg: {n_chunks} chunks in {len(batches)} length-bucketed batches, weights resident")
    for ids in batches[:3]:
        model.forward_ids(ids)
    torch.cuda.synchronize()
    log("encoder warm-up done")
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for ids in batches:
        model.forward_ids(ids)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    log(f"encoder timed: {dt:.3f} s")
    flops = float(sum(drv.flops_per_chunk(int(n), cfg) for n in lengths))
    padded_tokens = int(sum(int(b.numel()) for b in batches))
    res = {"metric": "chunks embedded/s (UniXcoder-geometry bf16 HIP encoder)", "value": world * n_chunks / dt, "unit": "chunks/s",
           "chunks_per_gpu": int(n_chunks), "seconds": dt, "mean_tokens": float(lengths.mean()), "true_tokens": int(lengths.sum()),
           "padded_tokens": padded_tokens, "batches": len(batches), "dtype": "bf16 weights/activations, f32 accumulate/LN/softmax",
           "data": "synthetic ids + seeded random RoBERTa-base-geometry weights (no checkpoint offline)",
           "roofline": {"bound": "mfma", "achieved": flops / dt / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": flops / dt / 1e12 / MFMA_BF16_PEAK_TFLOPS, "algorithmic_flops": flops}}
    if cpu:
        from oracle import encoder as orc
        ocfg = orc.EncoderConfig()
        w = orc.random_weights(ocfg, 23)
        torch.set_num_threads(host_threads())
        sample = lengths[:200]      # ~12 s of host work
        t0 = time.perf_counter()
        for n in sample:   # single-text calls, as the reference effectively issues them
            orc.forward(w, ocfg, orc.synthetic_ids(ocfg, [int(n)], 1))
        cdt = time.perf_counter() - t0
        log(f"encoder CPU baseline done: {cdt:.1f} s")
        res["cpu_baseline"] = {"value": len(sample) / cdt, "unit": "chunks/s", "cores": host_threads(), "kind": "port",
                               "sample": f"oracle/encoder.py torch-fp32 forward, {len(sample)} single-text calls "
                                         f"(mean {float(sample.mean()):.0f} tokens), {cdt:.1f} s"}
    return res


def cpu_baseline(np, B, K, D):
    """Exact cosine top-k on the host cores with the oracle's BLAS scan (oracle/search.py:search_blas):
    the stand-in for "Qdrant exact scan" (BASELINE.md section 3).  Bounded sample: 1M rows, ~10 s."""
    from oracle import search as orc
    from threadpoolctl import threadpool_limits
    rows = 1_000_000
    log("search CPU baseline: generating 1M-row sample")
    rng = np.random.default_rng(20251226)
    x = rng.standard_normal((rows, D), dtype=np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    q = orc.preprocess(np.random.default_rng(7).standard_normal((B, D)).astype(np.float32))
    with threadpool_limits(limits=host_threads()):
        orc.search_blas(x, q, K)
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 10.0:
            orc.search_blas(x, q, K)
            reps += 1
        dt = (time.perf_counter() - t0) / reps
    log("search CPU baseline done")
    return {"value": B / dt * (rows / 1e7), "unit": "queries/s per 10M-row shard (row.query pairs/s / 1e7)",
            "cores": host_threads(), "kind": "port",
            "sample": f"oracle BLAS exact scan (f32 sgemm + argpartition), {rows} rows x {B} queries top-{K}, "
                      f"{reps} batches of {dt * 1e3:.0f} ms, scaled to 10M rows"}


if __name__ == "__main__":
    main()
