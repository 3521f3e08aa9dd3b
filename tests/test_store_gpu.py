"""The QdrantManager-surface scenarios on the real HIP index (f32 store: ids and scores equal the f32 oracle)."""
import asyncio

import pytest

pytestmark = pytest.mark.gpu


def test_store_scenarios_on_hip_index(gpu):
    import coderag_amd  # noqa: F401
    from coderag_amd.store import HipVectorStore
    from tests.store_scenarios import run_store_scenarios
    asyncio.run(run_store_scenarios(HipVectorStore(dim=768, dtype="f32", initial_capacity=64, device=0)))


def test_reference_test_database_scenario_dim_1536(gpu, monkeypatch):
    """/root/reference/tests/test_database.py:64-124 on the HIP store at the reference's default EMBEDDING_DIMENSIONS = 1536
    (the store takes its dimension from the settings exactly as QdrantManager does: client.py:29)."""
    import coderag_amd  # noqa: F401
    from coderag_amd.store import CollectionName, QdrantManager
    from tests.store_scenarios import run_reference_database_scenario
    monkeypatch.delenv("EMBEDDING_DIMENSIONS", raising=False)
    monkeypatch.delenv("EMBEDDING_PROVIDER", raising=False)
    monkeypatch.setenv("CODERAG_HIP_STORE_DTYPE", "f32")
    for dtype in ("f32", "bf16"):
        monkeypatch.setenv("CODERAG_HIP_STORE_DTYPE", dtype)
        if dtype == "bf16":
            continue      # (scores of the bf16 store are checked against the bf16 oracle in the search tests)
        asyncio.run(run_reference_database_scenario(QdrantManager(), CollectionName))


def test_golden_search_cases_on_hip_index(gpu):
    import numpy as np
    import coderag_amd  # noqa: F401
    from coderag_amd import ffi
    from tests.search_cases import CASES, make_case
    gold = np.load(__file__.rsplit("/", 1)[0] + "/golden/search_cases.npz")
    for name in CASES:
        c = make_case(name)
        for bf16 in (False, True):
            tag = f"{name}/{'bf16' if bf16 else 'f32'}"
            ncols = 0 if c.get("codes") is None else c["codes"].shape[1]
            idx = ffi.Index(768, ffi.DTYPE_BF16 if bf16 else ffi.DTYPE_F32, capacity_rows=max(64, len(c["x"])), n_code_cols=ncols)
            idx.append(c["x"], c.get("codes"))
            if c.get("alive") is not None:
                idx.tombstone(np.flatnonzero(c["alive"] == 0))
            s, r = idx.search(c["q"], c["k"], filters=c.get("filters"))
            assert np.array_equal(r, gold[f"{tag}/rows"]), tag
            assert np.array_equal(s.view(np.uint32), gold[f"{tag}/scores"].view(np.uint32)), tag
            idx.close()


def test_store_scenarios_with_three_shards_on_hip_index(gpu):
    """HipVectorStore(shards=3), all shards on device 0 (what a one-GPU box can run): per-shard crh_search with row_base ->
    crh_merge_topk_strided on the device; filters / deletes / update check fan out; same scenarios, same answers."""
    import coderag_amd  # noqa: F401
    from coderag_amd.store import CollectionName, HipVectorStore, QdrantManager
    from tests.store_scenarios import run_reference_database_scenario, run_store_scenarios
    asyncio.run(run_store_scenarios(HipVectorStore(dim=768, dtype="f32", initial_capacity=64, device=0, shards=3)))
    asyncio.run(run_reference_database_scenario(QdrantManager(dim=1536, dtype="f32", shards=3), CollectionName))


def test_sharded_hip_store_equals_the_unsharded_one(gpu, monkeypatch):
    """The life of a collection of tests/test_store_sharded.py (bulk upsert in blocks over the shards, filters, replace, the
    reference's delete-then-reinsert re-index, raw-client delete, auto + explicit compaction, snapshot) on the real index with
    1, 2 and 3 shards: identical ids, f32 score bits and payloads throughout."""
    import coderag_amd  # noqa: F401
    from coderag_amd import shards as shards_mod
    from coderag_amd.store import HipVectorStore
    from tests.test_store_sharded import drive
    orig = shards_mod.ShardSet.__init__

    def small_blocks(self, *a, **kw):
        kw["block"] = 64
        orig(self, *a, **kw)
    monkeypatch.setattr(shards_mod.ShardSet, "__init__", small_blocks)
    logs = {}
    for ns in (1, 2, 3):
        logs[ns] = []
        asyncio.run(drive(HipVectorStore(dim=768, dtype="f32", initial_capacity=64, device=0, shards=ns, compact_dead_fraction=0.2,
                                         compact_min_dead=50), logs[ns]))
    assert logs[1] == logs[2] == logs[3] and len(logs[1]) > 25


def test_sharded_device_rerank_equals_the_host_ranker(gpu):
    """search_rerank_batch over three shards: per-shard scans -> device merge -> side columns gathered per shard and summed ->
    crh_rerank_vector; equal to the host HybridRanker on the same hits, and to the unsharded store."""
    import numpy as np
    import coderag_amd  # noqa: F401
    from coderag_amd.engine_helpers import search_and_rank_batch_device
    from coderag_amd.query_types import ExtractedEntity, QueryIntent, QueryPlan
    from coderag_amd.ranking import HybridRanker
    from coderag_amd.ranking.device import DeviceReranker
    from coderag_amd.store import HipVectorStore
    rng = np.random.default_rng(3)
    n = 700
    vecs = rng.standard_normal((n, 768)).astype(np.float32)
    pay = [{"file_path": f"/p/f{i % 40}.py", "entity_type": "function", "entity_name": f"fn_{i % 90}", "language": "python", "start_line": i % 17,
            "end_line": i % 17 + 5, "content": "x" * int(rng.integers(0, 3000)), "graph_node_id": None if i % 3 else f"m.fn_{i % 90}",
            "content_hash": "h", "project_name": "p"} for i in range(n)]
    qs = rng.standard_normal((8, 768)).astype(np.float32)
    plans = [QueryPlan(f"q{i}", list(QueryIntent)[i % len(QueryIntent)], entities=[ExtractedEntity(f"fn_{7 * i % 90}"), ExtractedEntity("fn")]) for i in range(8)]
    out = {}
    for ns in (1, 3):
        async def go(ns=ns):
            async with HipVectorStore(dim=768, dtype="f32", initial_capacity=64, device=0, shards=ns) as s:
                await s.create_collections()
                await s.upsert("code_chunks", [f"id{i}" for i in range(n)], vecs, pay)
                await s.set_graph_degrees("code_chunks", {f"m.fn_{j}": j for j in range(0, 90, 2)})
                return await search_and_rank_batch_device(s, DeviceReranker(device=0), HybridRanker(), qs, plans, limit=20)
        ranked = asyncio.run(go())
        out[ns] = [[(r.file_path, r.entity_name, r.start_line, r.final_score, r.source, tuple(sorted(r.signal_scores.items()))) for r in per] for per in ranked]
    assert out[1] == out[3] and all(len(per) > 0 for per in out[1])


def test_searches_beside_an_indexing_run_use_the_stores_own_priority_stream(gpu):
    """The reference serves queries while it indexes (one process: the encoder on the provider's worker thread,
    providers/unixcoder_provider.py:260; the store behind its own).  With both on the device's default stream a search queues
    behind every launch of the forward that is under way; the store therefore runs its device work on its own high-priority
    stream (`stream="priority"`, the default).  Same results either way; beside a long embedding call the searches of the
    priority store take milliseconds, those of a `stream="default"` store wait for whole forwards."""
    import time
    import numpy as np
    import torch
    import coderag_amd  # noqa: F401
    from coderag_amd.providers import HipUniXcoderProvider, ProviderConfig
    from coderag_amd.store import HipVectorStore
    provider = HipUniXcoderProvider(ProviderConfig(provider="unixcoder-hip", model="synthetic", extra={"synthetic_weights": 7}))   # 12 layers
    rng = np.random.default_rng(13)
    n = 300_000
    vecs = rng.standard_normal((n, 768)).astype(np.float32)
    pay = [{"file_path": f"/p/f{i % 100}.py", "entity_name": f"fn_{i}", "content": "x"} for i in range(n)]
    ids = [f"id{i}" for i in range(n)]
    texts = [("def f_%d(x):\n    return x + %d\n" % (i, i)) * 24 for i in range(6000)]          # ~400 tokens each: a few hundred ms of forwards
    queries = vecs[rng.choice(n, 24, replace=False)]

    async def run(mode):
        async with HipVectorStore(dim=768, dtype="bf16", initial_capacity=n, stream=mode) as store:
            await store.create_collections()
            for a in range(0, n, 100_000):
                await store.upsert("code_chunks", ids[a:a + 100_000], vecs[a:a + 100_000], pay[a:a + 100_000])
            await store.search("code_chunks", queries[0].tolist(), limit=5)
            alone = []
            for q in queries[:8]:
                t0 = time.perf_counter()
                await store.search("code_chunks", q.tolist(), limit=5)
                alone.append((time.perf_counter() - t0) * 1e3)
            emb = asyncio.ensure_future(provider.embed_batch(texts, batch_size=len(texts)))
            await asyncio.sleep(0.05)                                   # the embedding call is on the device by now
            beside, hits = [], []
            while not emb.done() and len(beside) < 200:
                q = queries[len(beside) % len(queries)]
                t0 = time.perf_counter()
                hits.append([(h["id"], h["score"]) for h in await store.search("code_chunks", q.tolist(), limit=5)])
                beside.append((time.perf_counter() - t0) * 1e3)
            await emb
            ref = [[(h["id"], h["score"]) for h in await store.search("code_chunks", queries[i % len(queries)].tolist(), limit=5)] for i in range(len(hits))]
            assert hits == ref and all(h[0][1] > 0.99 for h in hits)    # the same hits, the stored row itself first
            return float(np.median(alone)), float(np.median(beside)) if beside else None, max(beside) if beside else None, len(beside)
    asyncio.run(provider.embed_batch(texts[:512], batch_size=512))     # load + warm the encoder
    out = {mode: asyncio.run(run(mode)) for mode in ("default", "priority")}
    print({k: tuple(round(x, 2) if isinstance(x, float) else x for x in v) for k, v in out.items()})
    assert out["priority"][3] >= 5, "the embedding call ended before any search ran beside it"
    assert out["priority"][1] <= 5.0, out                               # milliseconds, not forwards
