#!/usr/bin/env python3
"""Host memory of the store's tables PER RANK with one process per shard (backend "dist"): ids stay replicated (16 bytes per
slot), the coded payload columns too, the payload TEXT (content: ~400 characters per chunk here) lives with the rank that owns
the row.  Runs on CPU: `world` gloo ranks, the device index replaced by a stub that keeps row counts only (nothing is searched),
`rows` synthetic chunks upserted through HipVectorStore.upsert(vectors=None, texts=..., embed=...), every rank embedding only
its own share (the stub embedder counts the texts it is given).
python tools/sharded_ingest_bytes.py [rows] [world] -> one JSON line (rank 0)."""
import asyncio
import json
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch.multiprocessing as mp


class StubIndex:
    """Row bookkeeping of ffi.Index, no vectors: enough for upserts (this tool measures the HOST tables)."""

    def __init__(self, dim=768, dtype=1, capacity_rows=65536, n_code_cols=0, device=0):
        self.dim, self.dtype, self.capacity_rows, self.n = dim, dtype, capacity_rows, 0

    def reserve(self, capacity_rows):
        self.capacity_rows = max(self.capacity_rows, capacity_rows)

    def append(self, vecs, codes=None, stream=0, preprocessed=False):
        first = self.n
        self.n += len(vecs)
        return first

    def count(self):
        return self.n, self.n

    def tombstone(self, rows):
        pass

    def close(self):
        pass


def uid(i: int) -> str:
    h = f"{(i * 0x9E3779B97F4A7C15 + 0x1234567) & ((1 << 128) - 1):032x}"
    return f"{h[:8]}-{h[8:12]}-{h[12:16]}-{h[16:20]}-{h[20:]}"


def worker(rank, world, port, rows, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import coderag_amd  # noqa: F401
    from coderag_amd import ffi
    from coderag_amd.store import HipVectorStore
    ffi.Index = StubIndex
    ffi.lib = lambda: object()
    ffi.device_count = lambda: 1
    ffi.device_info = lambda d=0: {"name": "stub", "arch": "gfx950", "hbm_bytes": 0, "cu_count": 256}
    body = "def f(x):\n    return x  # " + "lorem ipsum " * 31
    seen = [0]

    def embed(texts):
        seen[0] += len(texts)
        return np.zeros((len(texts), 768), np.float32)

    async def go():
        async with HipVectorStore(dim=768, dtype="bf16", initial_capacity=rows + 64, shards=world) as s:
            await s.create_collections()
            batch = 50_000
            for a in range(0, rows, batch):
                m = min(batch, rows - a)
                texts = [f"{body}{a + i}" for i in range(m)]
                pay = [{"file_path": f"/p/f{(a + i) % 1000}.py", "entity_type": "function", "entity_name": f"fn_{a + i}", "language": "python",
                        "start_line": i, "end_line": i + 9, "content": t, "graph_node_id": f"m.fn_{a + i}", "content_hash": "h", "project_name": "p"}
                       for i, t in enumerate(texts)]
                await s.upsert("code_chunks", [uid(a + i) for i in range(m)], None, pay, texts=texts, embed=embed)
            col = s._collections["code_chunks"]
            return {"rank": rank, "texts_embedded": seen[0], "ids_bytes": int(col.ids.nbytes()), "payload_bytes": int(col.payloads.nbytes()),
                    "content_blob_bytes": len(col.payloads.cols["content"].blob), "shard_rows": list(col.shards.rows)}
    rec = asyncio.run(go())
    if world > 1:
        allr = [None] * world
        dist.all_gather_object(allr, rec)
        dist.destroy_process_group()
    else:
        allr = [rec]
    if rank == 0:
        json.dump(allr, open(out, "w"))


def run(rows, world):
    import tempfile
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    out = os.path.join(tempfile.mkdtemp(), "out.json")
    if world > 1:
        mp.spawn(worker, args=(world, port, rows, out), nprocs=world, join=True)
    else:
        worker(0, 1, port, rows, out)
    return json.load(open(out))


if __name__ == "__main__":
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    one = run(rows, 1)[0]
    many = run(rows, world)
    print(json.dumps({"rows": rows, "world": world, "unsharded": one, "per_rank": many,
                      "host_table_bytes_per_rank_over_unsharded": [round((r["ids_bytes"] + r["payload_bytes"]) / (one["ids_bytes"] + one["payload_bytes"]), 3) for r in many],
                      "texts_embedded_per_rank": [r["texts_embedded"] for r in many]}))
