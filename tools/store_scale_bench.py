#!/usr/bin/env python3
"""HipVectorStore's host side at the sizes its device side is built for: upsert rate with ndarray / CUDA
vectors and payload dictionaries, delete-by-file_path of 1 % of the rows, the update check, a filtered search -- on N synthetic
chunks (default 5M, bf16 store).  python tools/store_scale_bench.py [rows] -> one JSON line."""
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


async def main():
    s = HipVectorStore(dim=768, dtype="bf16", initial_capacity=rows)
    await s.connect()
    await s.create_collections()
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    t_up = t_pay = 0.0
    for r0 in range(0, rows, batch):
        m = min(batch, rows - r0)
        t0 = time.perf_counter()
        ids = [f"{i:032x}" for i in range(r0, r0 + m)]
        payloads = [{"file_path": f"/repo/src/f{i % n_files}.py", "entity_type": "function", "entity_name": f"fn_{i}", "language": "python",
                     "start_line": i % 900, "end_line": i % 900 + 20, "content": "def f(): pass", "graph_node_id": None,
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
    await s.close()
    print(json.dumps(out))

asyncio.run(main())
