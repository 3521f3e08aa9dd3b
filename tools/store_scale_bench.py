#!/usr/bin/env python3
"""HipVectorStore's host side at the sizes its device side is built for: upsert rate with ndarray / CUDA vectors and payload
dictionaries (uuid4-shaped ids, 400-character contents), delete-by-file_path of 1 % of the rows, the update check, a filtered
search; then half of the files deleted -> batch-64 scan time before / after HipVectorStore.compact() (crh_index_compact + the
columnar host tables), snapshot save / load of the whole store, resident host memory -- on N synthetic chunks (default 5M,
bf16 store).  python tools/store_scale_bench.py [rows] [snapshot dir] -> one JSON line."""
import asyncio
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import coderag_amd  # noqa: F401
from coderag_amd.store import HipVectorStore

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
batch = 100_000
n_files = 100                                   # one file = 1 % of the rows
dev = torch.device("cuda:0")


body = "def f(x):\n    return x  # " + "lorem ipsum " * 31            # ~400 characters per chunk


def uid(i: int) -> str:
    h = f"{(i * 0x9E3779B97F4A7C15 + 0x1234567) & ((1 << 128) - 1):032x}"
    return f"{h[:8]}-{h[8:12]}-{h[12:16]}-{h[16:20]}-{h[20:]}"


def rss_gb() -> float:
    import psutil
    return psutil.Process().memory_info().rss / 2**30


async def scan_ms(s, steps=10):
    """Device time of a 64-query batch top-100 through the collection's index (what the headline bench measures)."""
    idx = s._col("code_chunks").index
    qd = torch.randn((64, 768), device=dev)
    out_s = torch.empty((64, 100), dtype=torch.float32, device=dev)
    out_r = torch.empty((64, 100), dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        idx.search(qd, 100, out_scores=out_s, out_rows=out_r, stream=st)
    idx.search_finish(st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        idx.search(qd, 100, out_scores=out_s, out_rows=out_r, stream=st)
    e1.record()
    idx.search_finish(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps, idx.stats()["rows"] // (steps + 0)


async def main():
    s = HipVectorStore(dim=768, dtype="bf16", initial_capacity=rows + batch, compact_dead_fraction=0.0)      # (compaction is timed explicitly below)
    await s.connect()
    await s.create_collections()
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    t_up = t_pay = 0.0
    for r0 in range(0, rows, batch):
        m = min(batch, rows - r0)
        t0 = time.perf_counter()
        ids = [uid(i) for i in range(r0, r0 + m)]
        payloads = [{"file_path": f"/repo/src/f{i % n_files}.py", "entity_type": "function", "entity_name": f"fn_{i % 50000}", "language": "python",
                     "start_line": i % 900, "end_line": i % 900 + 20, "content": body + str(i), "graph_node_id": None,
                     "content_hash": f"h{i % n_files}", "project_name": "bench"} for i in range(r0, r0 + m)]
        t_pay += time.perf_counter() - t0
        x = torch.randn((m, 768), generator=gen, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        await s.upsert("code_chunks", ids, x, payloads)
        t_up += time.perf_counter() - t0
        if r0 % (10 * batch) == 0:
            print(f"[store bench] {r0 + m} rows, {(r0 + m) / t_up:.0f} rows/s so far", file=sys.stderr, flush=True)
    # the same rate question for host arrays (one batch, appended and deleted again by project)
    xh = np.random.default_rng(1).standard_normal((batch, 768)).astype(np.float32)
    pl = [{"file_path": f"/other/g{i % 10}.py", "entity_name": f"g{i}", "language": "go", "project_name": "tmp"} for i in range(batch)]
    t0 = time.perf_counter()
    await s.upsert("code_chunks", [f"x{i}" for i in range(batch)], xh, pl)
    t_np = time.perf_counter() - t0
    q = np.random.default_rng(7).standard_normal(768).astype(np.float32).tolist()
    await s.search("code_chunks", q, limit=10)
    t0 = time.perf_counter()
    hits = await s.search("code_chunks", q, limit=10, filters={"file_path": "/repo/src/f7.py"})
    t_fsearch = time.perf_counter() - t0
    t0 = time.perf_counter()
    fresh = await s.file_needs_update("code_chunks", "/repo/src/f7.py", "h7")
    t_check = time.perf_counter() - t0
    before = (await s.get_collection_info("code_chunks")).points_count
    t0 = time.perf_counter()
    await s.delete("code_chunks", {"file_path": "/repo/src/f7.py"})
    t_del = time.perf_counter() - t0
    after = (await s.get_collection_info("code_chunks")).points_count
    gone = await s.search("code_chunks", q, limit=10, filters={"file_path": "/repo/src/f7.py"})
    t0 = time.perf_counter()
    await s.delete("code_chunks", {"project_name": "tmp"})
    t_del2 = time.perf_counter() - t0
    out = {"rows": rows, "upsert_rows_per_s_cuda_vectors": rows / t_up, "upsert_seconds": t_up, "payload_build_seconds_not_counted": t_pay,
           "upsert_rows_per_s_ndarray_batch": batch / t_np, "filtered_search_ms": t_fsearch * 1e3, "file_needs_update_ms": t_check * 1e3,
           "file_needs_update_answer": fresh, "delete_by_file_path_ms": t_del * 1e3, "rows_deleted": before - after,
           "deleted_fraction": (before - after) / before, "hits_before_delete": len(hits), "hits_after_delete": len(gone),
           "delete_by_project_ms": t_del2 * 1e3, "points_left": (await s.get_collection_info("code_chunks")).points_count}
    col = s._col("code_chunks")
    out["host_tables_bytes"] = {"ids": col.ids.nbytes(), "payload_columns": col.payloads.nbytes()}
    out["rss_gb_after_build"] = rss_gb()
    # ---- snapshot of the whole store at full size (index image + id / payload tables), then restore
    snap = sys.argv[2] if len(sys.argv) > 2 else "/dev/shm/coderag_store_snap"
    probe_q = np.random.default_rng(9).standard_normal(768).astype(np.float32).tolist()
    live = await s.search("code_chunks", probe_q, limit=20)
    t0 = time.perf_counter()
    await s.save(snap)
    t_save = time.perf_counter() - t0
    size = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(snap) for f in fs)
    t0 = time.perf_counter()
    await s.load(snap)
    t_load = time.perf_counter() - t0
    out["snapshot_full"] = {"rows": (await s.get_collection_info("code_chunks")).config["rows_appended"], "save_seconds": t_save, "load_seconds": t_load,
                            "bytes": size, "directory": snap, "same_hits_after_restore": (await s.search("code_chunks", probe_q, limit=20)) == live}
    out["rss_gb_after_restore"] = rss_gb()
    # ---- half of the files deleted (the state four re-index runs without compaction would leave is worse): scan before / after
    ms_full, _ = await scan_ms(s)
    for f in range(0, n_files, 2):
        await s.delete("code_chunks", {"file_path": f"/repo/src/f{f}.py"})
    info = await s.get_collection_info("code_chunks")
    ms_dead, _ = await scan_ms(s)
    keep_q = np.random.default_rng(9).standard_normal(768).astype(np.float32).tolist()
    before = await s.search("code_chunks", keep_q, limit=20)
    t0 = time.perf_counter()
    reclaimed = await s.compact("code_chunks")
    t_compact = time.perf_counter() - t0
    after = await s.search("code_chunks", keep_q, limit=20)
    ms_compact, _ = await scan_ms(s)
    info2 = await s.get_collection_info("code_chunks")
    out["compaction"] = {"rows_before": info.config["rows_appended"], "alive": info.points_count, "rows_reclaimed": reclaimed, "seconds": t_compact,
                         "scan_ms_all_alive": ms_full, "scan_ms_half_dead": ms_dead, "scan_ms_after_compaction": ms_compact,
                         "rows_after": info2.config["rows_appended"], "same_hits_before_and_after": before == after}
    # ---- and the compacted store
    t0 = time.perf_counter()
    await s.save(snap)
    t_save = time.perf_counter() - t0
    size = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(snap) for f in fs)
    t0 = time.perf_counter()
    await s.load(snap)
    t_load = time.perf_counter() - t0
    again = await s.search("code_chunks", keep_q, limit=20)
    out["snapshot_compacted"] = {"rows": info2.config["rows_appended"], "save_seconds": t_save, "load_seconds": t_load, "bytes": size, "directory": snap,
                       "same_hits_after_restore": again == after}
    out["rss_gb_end"] = rss_gb()
    import shutil
    shutil.rmtree(snap, ignore_errors=True)
    await s.close()
    print(json.dumps(out))

asyncio.run(main())
