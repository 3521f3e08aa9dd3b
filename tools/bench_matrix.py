#!/usr/bin/env python3
"""Measurement matrix of SURVEY.md section 8(d) beyond the headline line of bench.py (one GPU):

  search  : bf16 / f32 store, Gaussian and clustered corpus, one filtered run (3 uniform `language` codes),
            query-count and k sweeps -- 10 warm-up + 50 timed batches each, per-batch device time (median/p10/p90),
            HIP-event time of the scan kernel (parity of the same corpus kinds -- Gaussian, clustered, filtered, f32 -- is
            the GPU test suite's job: tests/test_search_gpu.py);
  encoder : fixed-shape sweeps L in {128, 512} x B in {16, 64, 256} (median of 20 forwards).

    python tools/bench_matrix.py --out gpurun_out/matrix/matrix.json [--rows 10000000]

"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
_T0 = time.perf_counter()


def log(msg):
    print(f"[matrix +{time.perf_counter() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def pct(a):
    import numpy as np
    return {"median": float(np.median(a)), "p10": float(np.percentile(a, 10)), "p90": float(np.percentile(a, 90))}


def build_corpus(torch, ffi, dev, rows, dtype, kind, codes_cols, check_rows):
    """Gaussian: rows ~ N(0,I).  Clustered: 1000 unit centres + 0.7 * N(0,I)/sqrt(D) noise (normalised on insert)."""
    D = 768
    idx = ffi.Index(D, dtype, capacity_rows=rows, n_code_cols=codes_cols, device=dev.index)
    gen = torch.Generator(device=dev)
    gen.manual_seed(20251226)
    centres = None
    if kind == "clustered":
        centres = torch.randn((1000, D), generator=gen, device=dev)
        centres /= centres.norm(dim=1, keepdim=True)
    head = head_codes = None
    stream = torch.cuda.current_stream().cuda_stream
    block = 500_000
    for r0 in range(0, rows, block):
        m = min(block, rows - r0)
        xb = torch.randn((m, D), generator=gen, device=dev)
        if centres is not None:
            pick = torch.randint(0, 1000, (m,), generator=gen, device=dev)
            xb = centres[pick] + 0.7 * xb / (D ** 0.5)
        cb = None
        if codes_cols:
            cb = torch.randint(0, 3, (m, codes_cols), generator=gen, device=dev, dtype=torch.int32)
        idx.append(xb, codes=cb, stream=stream)
        if r0 == 0:
            head = xb[:check_rows].cpu().numpy()
            head_codes = cb[:check_rows].cpu().numpy() if cb is not None else None
        torch.cuda.synchronize()
        del xb
    return idx, head, head_codes, centres


def time_search(torch, idx, qd, k, filters, steps, warmup):
    nq = qd.shape[0]
    dev = qd.device
    stream = torch.cuda.current_stream().cuda_stream
    out_s = [torch.empty((nq, k), dtype=torch.float32, device=dev) for _ in range(4)]
    out_r = [torch.empty((nq, k), dtype=torch.int64, device=dev) for _ in range(4)]
    for i in range(warmup):
        idx.search(qd, k, filters=filters, out_scores=out_s[i % 4], out_rows=out_r[i % 4], stream=stream)
    idx.search_finish(stream)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    idx.set_profiling(True)
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(steps):
        idx.search(qd, k, filters=filters, out_scores=out_s[i % 4], out_rows=out_r[i % 4], stream=stream)
        ev[i + 1].record()
    idx.search_finish(stream)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    scan_ms, launches = idx.profile()
    idx.set_profiling(False)
    import numpy as np
    per = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(steps)])
    st = idx.stats()
    return {"nq": int(nq), "k": int(k), "steps": steps, "warmup": warmup, "wall_ms_per_call": wall * 1e3 / steps,
            "queries_per_s": nq * steps / wall, "call_ms_device": pct(per),
            "scan_kernel_ms": scan_ms / max(1, launches), "scan_launches_per_call": launches / steps,
            "candidates": st.get("candidates"), "fallback_used": st.get("fallback_used")}


def c5_leg(np, torch, ffi, dev, rows, steps, warmup):
    from types import SimpleNamespace as NS
    from coderag_amd.query_types import GraphContext, QueryIntent
    from coderag_amd.ranking import HybridRanker
    from coderag_amd.ranking.device import DeviceReranker, SideColumns, SIGNALS
    from coderag_amd.engine_helpers import centrality_candidates
    D, nq, k = 768, 64, 100
    idx, _, _, _ = build_corpus(torch, ffi, dev, rows, ffi.DTYPE_BF16, "gaussian", 0, 1)
    vocab = [f"fn_{i}" for i in range(2000)] + ["UserRepository", "verify_password", "parse_file", ""]
    r99, r98 = np.random.default_rng(99), np.random.default_rng(98)
    name_id = r99.integers(0, len(vocab), rows)
    names = np.zeros((len(vocab), 64), np.uint8)
    nlen = np.zeros(len(vocab), np.int32)
    for i, v in enumerate(vocab):
        b = v.lower().encode()
        names[i, :len(b)] = np.frombuffer(b, np.uint8)
        nlen[i] = len(b)
    file_code = (np.arange(rows) // 7 + 1).astype(np.int32)                     # 7 chunks per file
    side = SideColumns.from_arrays(0, content_len=np.clip(np.round(np.exp(r99.normal(np.log(400.0), 1.0, rows))), 0, 20000),
                                   degree=np.minimum(r98.zipf(1.6, rows), 500) - 1, file_code=file_code,
                                   key_code=np.arange(1, rows + 1), node_code=name_id + 1, name_len=nlen[name_id], name=names[name_id])
    intents = [i.value for i in QueryIntent]
    plans = [NS(primary_intent=intents[q % len(intents)], entities=[NS(name=vocab[(37 * q) % 2000]), NS(name="Repository")]) for q in range(nq)]
    qd = torch.from_numpy(np.random.default_rng(7).standard_normal((nq, D)).astype(np.float32)).to(dev)
    stream = torch.cuda.current_stream().cuda_stream
    s = torch.empty((nq, k), dtype=torch.float32, device=dev)
    r = torch.empty((nq, k), dtype=torch.int64, device=dev)
    rr = DeviceReranker()

    def once():
        idx.search(qd, k, out_scores=s, out_rows=r, stream=stream)
        idx.search_finish(stream)
        return rr.rank(s, r, side.gather(r, stream=stream), plans, stream=stream)
    for _ in range(warmup):
        out = once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = once()                       # (rank() ends with the D2H copy of the survivors: the step is complete)
    wall = (time.perf_counter() - t0) / steps
    # re-rank alone (gather + kernel + D2H), same candidates
    t0 = time.perf_counter()
    for _ in range(steps):
        rr.rank(s, r, side.gather(r, stream=stream), plans, stream=stream)
    rerank = (time.perf_counter() - t0) / steps
    # parity + host cost: HybridRanker over hit dicts rebuilt from the side arrays for these candidates
    rows_h, scores_h = r.cpu().numpy(), s.cpu().numpy()
    host = HybridRanker()
    cl, dg = np.asarray(side._host["content_len"]), np.asarray(side._host["degree"])
    ok, t_host = True, 0.0
    for q in range(nq):
        hits = [{"score": float(sc), "file_path": f"f{file_code[ro]}", "entity_name": vocab[name_id[ro]], "entity_type": "function",
                 "graph_node_id": None, "content": "x" * int(cl[ro]) if cl[ro] else None, "start_line": int(ro), "end_line": 0}
                for ro, sc in zip(rows_h[q], scores_h[q]) if ro >= 0]
        cand = centrality_candidates(GraphContext(), hits)
        deg_of = {vocab[name_id[ro]]: int(dg[ro]) for ro in rows_h[q][:5] if ro >= 0}
        table = {n: {"total_degree": deg_of[n]} for n in cand if deg_of.get(n, -1) >= 0}
        t1 = time.perf_counter()
        want = host.rank_results(plans[q], GraphContext(), hits, table)
        t_host += time.perf_counter() - t1
        got = DeviceReranker.materialise(out, q, hits)
        ok = ok and len(got) == len(want) and all(g.final_score == w.final_score and g.entity_name == w.entity_name and g.start_line == w.start_line
                                                   and [g.signal_scores[n] for n in SIGNALS] == [w.signal_scores[n] for n in SIGNALS]
                                                   for g, w in zip(got, want))
    idx.close()
    log(f"  C5: scan+rerank {wall * 1e3:.3f} ms/batch, rerank part {rerank * 1e3:.3f} ms, host ranker {t_host * 1e3:.1f} ms, parity {ok}")
    return {"workload": f"{rows} x {D} bf16, {nq} queries, top-{k} -> hybrid re-rank (vector branch) -> <= 50 per query",
            "scan_plus_rerank_ms_per_batch": wall * 1e3, "rerank_ms_per_batch": rerank * 1e3, "queries_per_s": nq / wall,
            "host_hybrid_ranker_ms_per_batch": t_host * 1e3, "identical_to_host_ranker": bool(ok)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--check-rows", type=int, default=200_000)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "matrix", "matrix.json"))
    ap.add_argument("--skip-encoder", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import coderag_amd  # noqa: F401
    from coderag_amd import ffi

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    D = 768
    qs = np.random.default_rng(7).standard_normal((1024, D)).astype(np.float32)
    res = {"rows": args.rows, "dim": D, "device": ffi.device_info(0), "search": [], "encoder": []}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)

    def flush():
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)

    variants = [("bf16", "gaussian", 0), ("f32", "gaussian", 0), ("bf16", "clustered", 0), ("bf16", "gaussian", 1)]
    for dt_name, kind, ncols in variants:
        dtype = ffi.DTYPE_BF16 if dt_name == "bf16" else ffi.DTYPE_F32
        idx, head, head_codes, centres = build_corpus(torch, ffi, dev, args.rows, dtype, kind, ncols, args.check_rows)
        log(f"corpus {dt_name}/{kind}/codes={ncols} resident")
        if centres is not None:   # queries near the data: a centre + the same noise model
            pick = np.random.default_rng(8).integers(0, 1000, 1024)
            q_all = (centres.cpu().numpy()[pick] + 0.7 * qs / np.sqrt(D)).astype(np.float32)
        else:
            q_all = qs
        filt = [(0, 1)] if ncols else None
        runs = [(64, 100)]
        if dt_name == "bf16" and kind == "gaussian" and not ncols:
            runs += [(1, 100), (8, 100), (256, 100), (1024, 100), (64, 10), (64, 1000)]
        for nq, k in runs:
            qd = torch.from_numpy(np.ascontiguousarray(q_all[:nq])).to(dev)
            steps = args.steps if nq <= 256 else max(5, args.steps // 5)
            r = time_search(torch, idx, qd, k, filt, steps, args.warmup if nq <= 256 else 2)
            r.update(store=dt_name, corpus=kind, filter=("language == code 1 of 3 (uniform)" if ncols else None))
            if nq == 64 and k == 100:
                alg = float(args.rows) * D * 2
                r["scan_GBps"] = alg / (r["scan_kernel_ms"] * 1e-3) / 1e9
                r["scan_frac_of_8TBps"] = r["scan_GBps"] / 8000.0
            res["search"].append(r)
            log(f"  nq={nq} k={k}: {r['wall_ms_per_call']:.3f} ms/call, scan {r['scan_kernel_ms']:.3f} ms")
            flush()
        idx.close()
        del idx
        torch.cuda.empty_cache()
        flush()

    # ---- BASELINE config 5 on one GPU: scan + device hybrid re-rank of the top-100 of 64 queries, side arrays per SURVEY 8(d)
    # (content_len lognormal seed 99, degree Zipf seed 98), one plan per query cycling the intents
    try:
        res["c5"] = c5_leg(np, torch, ffi, dev, args.rows, args.steps, args.warmup)
    except Exception as e:   # the matrix's other rows stay useful
        res["c5"] = {"error": repr(e)}
    flush()

    if not args.skip_encoder:
        from coderag_amd import encoder as drv
        cfg = drv.EncoderConfig()
        model = drv.HipUniXcoder(drv.synthetic_weights(cfg, 23), cfg, drv.HashTokenizer(cfg.vocab_size), 0)
        rng = np.random.default_rng(1234)
        for L in (128, 512):
            for B in (16, 64, 256):
                ids = rng.integers(16, cfg.vocab_size, (B, L)).astype(np.int32)
                ids[:, 0], ids[:, 1], ids[:, 2], ids[:, -1] = 0, 5, 2, 2
                t = torch.from_numpy(ids).to(dev)
                for _ in range(3):
                    model.forward_ids(t)
                torch.cuda.synchronize()
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
                ev[0].record()
                for i in range(20):
                    model.forward_ids(t)
                    ev[i + 1].record()
                torch.cuda.synchronize()
                per = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(20)])
                fl = B * drv.flops_per_chunk(L, cfg)
                med = float(np.median(per))
                res["encoder"].append({"B": B, "L": L, "forward_ms": pct(per), "chunks_per_s": B / med * 1e3,
                                       "TFLOPs": fl / med / 1e9, "frac_of_2.5PF": fl / med / 1e9 / 2500.0})
                log(f"  encoder B={B} L={L}: {med:.3f} ms")
                flush()
    flush()
    print(json.dumps({"written": args.out, "search_rows": len(res["search"]), "encoder_rows": len(res["encoder"])}))


if __name__ == "__main__":
    main()
