#!/usr/bin/env python3
"""Many callers, one query each, through HipVectorStore.search (the shape of the reference's query traffic): throughput and
latency with and without the coalescing window.  python tools/serve_bench.py [rows] [callers]"""
import asyncio, os, sys, time, uuid
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import coderag_amd
from coderag_amd.store import HipVectorStore
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
callers = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
rng = np.random.default_rng(0)


async def main():
    for window in (-1.0, 0.0, 0.3):
        async with HipVectorStore(dim=768, dtype="bf16", initial_capacity=rows, search_window_ms=window) as s:
            await s.create_collections()
            for r0 in range(0, rows, 100_000):
                m = min(100_000, rows - r0)
                v = rng.standard_normal((m, 768)).astype(np.float32)
                await s.upsert("code_chunks", [str(uuid.UUID(int=r0 + i)) for i in range(m)], v,
                               [{"file_path": f"f{(r0 + i) % 5000}.py", "entity_name": f"e{r0 + i}", "language": "python"} for i in range(m)])
            qs = rng.standard_normal((callers, 768)).astype(np.float32).tolist()
            await asyncio.gather(*(s.search("code_chunks", q, limit=10) for q in qs[:64]))
            lat = []

            async def one(q):
                t0 = time.perf_counter()
                await s.search("code_chunks", q, limit=10)
                lat.append((time.perf_counter() - t0) * 1e3)
            s.search_passes = 0
            t0 = time.perf_counter()
            await asyncio.gather(*(one(q) for q in qs))
            dt = time.perf_counter() - t0
            one_t = time.perf_counter()
            await s.search("code_chunks", qs[0], limit=10)
            one_ms = (time.perf_counter() - one_t) * 1e3
            print(f"window {window} ms, {rows} rows: {callers} concurrent single-query searches in {dt * 1e3:.0f} ms = {callers / dt:.0f} queries/s, "
                  f"{s.search_passes} corpus passes, median latency {np.median(lat):.1f} ms; a lone search: {one_ms:.2f} ms", flush=True)



async def text_queries():
    """The reference's query path end to end (minus the LLM): VectorSearcher.search_code(text) = embed the query (provider,
    coalesced) + search (store, coalesced) + payload projection."""
    from coderag_amd.embedder import Embedder
    from coderag_amd.providers import HipUniXcoderProvider, ProviderConfig
    from coderag_amd.vector_search import VectorSearcher
    provider = HipUniXcoderProvider(ProviderConfig(provider="unixcoder-hip", model="synthetic", extra={"synthetic_weights": 5}))
    embedder = Embedder(provider_instance=provider)
    async with HipVectorStore(dim=768, dtype="bf16", initial_capacity=rows) as s:
        await s.create_collections()
        for r0 in range(0, rows, 100_000):
            m = min(100_000, rows - r0)
            v = rng.standard_normal((m, 768)).astype(np.float32)
            await s.upsert("code_chunks", [str(uuid.UUID(int=r0 + i)) for i in range(m)], v,
                           [{"file_path": f"f{(r0 + i) % 5000}.py", "entity_name": f"e{r0 + i}", "language": "python", "content": "x = 1"} for i in range(m)])
        searcher = VectorSearcher(qdrant=s, embedder=embedder)
        texts = [f"how does function number {i} parse the {i % 17} configuration file and return a value" for i in range(callers)]
        await asyncio.gather(*(searcher.search_code(t, limit=10) for t in texts[:64]))
        lat = []

        async def one(t):
            t0 = time.perf_counter()
            await searcher.search_code(t, limit=10)
            lat.append((time.perf_counter() - t0) * 1e3)
        t0 = time.perf_counter()
        await asyncio.gather(*(one(t) for t in texts))
        dt = time.perf_counter() - t0
        t1 = time.perf_counter()
        await searcher.search_code(texts[0], limit=10)
        lone = (time.perf_counter() - t1) * 1e3
        print(f"text queries, {rows} rows: {callers} concurrent search_code() in {dt * 1e3:.0f} ms = {callers / dt:.0f} queries/s, median latency "
              f"{np.median(lat):.1f} ms; a lone search_code(): {lone:.2f} ms", flush=True)

if len(sys.argv) > 3 and sys.argv[3] == "text":
    asyncio.run(text_queries())
else:
    asyncio.run(main())
