"""HipVectorStore's QdrantManager surface (src/lattice/embeddings/client.py:18-228) -- host-side behaviour, with the
device index replaced by the oracle-backed FakeIndex.  The same scenarios run against the real index in
tests/test_store_gpu.py."""
import asyncio

import numpy as np
import pytest

import coderag_amd  # noqa: F401
from coderag_amd import ffi, store as store_mod
from coderag_amd.errors import VectorStoreError
from tests.fake_index import FakeIndex
from tests.store_scenarios import run_store_scenarios


@pytest.fixture
def fake(monkeypatch):
    monkeypatch.setattr(ffi, "Index", FakeIndex)
    monkeypatch.setattr(ffi, "lib", lambda: object())
    monkeypatch.setattr(ffi, "device_count", lambda: 1)
    monkeypatch.setattr(ffi, "device_info", lambda d=0: {"name": "fake", "arch": "gfx950", "hbm_bytes": 0, "cu_count": 256})
    return FakeIndex


def test_store_scenarios_on_fake_index(fake):
    asyncio.run(run_store_scenarios(store_mod.HipVectorStore(dim=768, initial_capacity=64)))


def test_reference_test_database_scenario_dim_1536(fake, monkeypatch):
    """/root/reference/tests/test_database.py:64-124, host logic over the oracle-backed index (GPU tier: test_store_gpu.py)."""
    from tests.store_scenarios import run_reference_database_scenario
    monkeypatch.delenv("EMBEDDING_DIMENSIONS", raising=False)
    monkeypatch.delenv("EMBEDDING_PROVIDER", raising=False)
    asyncio.run(run_reference_database_scenario(store_mod.QdrantManager(), store_mod.CollectionName))


def test_client_before_connect_raises(fake):
    s = store_mod.HipVectorStore(dim=768)
    with pytest.raises(VectorStoreError, match="Client not connected"):
        _ = s.client

    async def go():
        with pytest.raises(VectorStoreError):
            await s.create_collections()
        assert await s.health_check() is False
        assert await s.file_needs_update("code_chunks", "a.py", "h") is True      # True on ANY error
    asyncio.run(go())


def test_connect_failure_is_wrapped(monkeypatch):
    monkeypatch.setattr(ffi, "device_count", lambda: 0)
    s = store_mod.HipVectorStore(dim=768)

    async def go():
        with pytest.raises(VectorStoreError, match="Failed to connect"):
            await s.connect()
    asyncio.run(go())


def test_dimension_defaults_follow_provider(monkeypatch):
    monkeypatch.setenv("EMBEDDING_PROVIDER", "unixcoder")
    assert store_mod.HipVectorStore()._dimensions == 768
    monkeypatch.setenv("EMBEDDING_PROVIDER", "openai")
    monkeypatch.setenv("EMBEDDING_DIMENSIONS", "1536")
    assert store_mod.HipVectorStore()._dimensions == 1536         # quirk Q3 reproduced unless dim= is passed
    assert store_mod.QdrantManager is store_mod.HipVectorStore
    assert [c.value for c in store_mod.CollectionName] == ["code_chunks", "summaries"]


def test_concurrent_searches_share_corpus_passes(fake):
    """Many callers, one query each (how the reference's query path arrives): what lands within the window, for the same
    collection and filter, is served by one pass; every caller gets exactly what a lone call returns (its own limit, its
    own filter), and a malformed query fails alone."""
    import uuid
    rng = np.random.default_rng(4)
    n = 500
    vecs = rng.standard_normal((n, 768)).astype(np.float32)
    pay = [{"file_path": f"f{i % 7}.py", "entity_name": f"e{i}", "language": ("python", "go")[i % 2], "content": "x"} for i in range(n)]
    qs = rng.standard_normal((90, 768)).astype(np.float32)
    limits = [int(rng.integers(1, 30)) for _ in range(90)]
    filts = [None if i % 3 else {"language": "go"} for i in range(90)]

    async def go(window_ms):
        async with store_mod.HipVectorStore(dim=768, initial_capacity=1024, search_window_ms=window_ms) as s:
            await s.create_collections()
            await s.upsert("code_chunks", [str(uuid.UUID(int=i)) for i in range(n)], vecs, pay)
            res = await asyncio.gather(*(s.search("code_chunks", qs[i].tolist(), limit=limits[i], filters=filts[i]) for i in range(90)))
            with pytest.raises(VectorStoreError):
                await s.search("code_chunks", [0.0] * 10, limit=3)
            return res, s.search_passes
    lone, passes_lone = asyncio.run(go(-1))
    together, passes_together = asyncio.run(go(0))
    assert passes_lone == 90 and passes_together <= 4
    assert [[(h["id"], h["score"]) for h in r] for r in together] == [[(h["id"], h["score"]) for h in r] for r in lone]
    assert all(len(r) == min(limits[i], n if filts[i] is None else n // 2) for i, r in enumerate(together))


def test_limit_above_the_index_maximum_fails_only_its_own_caller(fake):
    """A limit the index cannot serve (k > 1024) is refused before the query joins a coalesced pass: concurrent searches of
    other callers still succeed."""
    async def go():
        s = store_mod.HipVectorStore(dim=768, initial_capacity=64)
        async with s:
            await s.create_collections()
            rng = np.random.default_rng(0)
            vecs = rng.standard_normal((20, 768)).astype(np.float32)
            await s.upsert("code_chunks", [f"i{i}" for i in range(20)], vecs, [{"file_path": "a.py"} for _ in range(20)])
            res = await asyncio.gather(s.search("code_chunks", vecs[0].tolist(), limit=3),
                                       s.search("code_chunks", vecs[1].tolist(), limit=5000),
                                       s.search("code_chunks", vecs[2].tolist(), limit=2), return_exceptions=True)
            assert len(res[0]) == 3 and len(res[2]) == 2 and res[0][0]["id"] == "i0"
            assert isinstance(res[1], VectorStoreError) and "1024" in str(res[1])
    asyncio.run(go())


def test_raw_client_matchtext_resolves_through_the_code_dictionaries(fake):
    """projects/cleanup.py:41-61: count / delete with MatchText on file_path -- every code whose value contains the text."""
    from types import SimpleNamespace as M

    async def go():
        s = store_mod.HipVectorStore(dim=768, initial_capacity=64)
        async with s:
            await s.create_collections()
            rng = np.random.default_rng(1)
            files = [f"/w/projA/f{i % 5}.py" if i % 2 else f"/w/projB/g{i % 3}.py" for i in range(60)]
            await s.upsert("code_chunks", [f"i{i}" for i in range(60)], rng.standard_normal((60, 768)).astype(np.float32),
                           [{"file_path": f, "language": "python" if i % 3 else "go"} for i, f in enumerate(files)])
            flt = M(must=[M(key="file_path", match=M(text="/projA/"))])
            assert (await s.client.count("code_chunks", count_filter=flt)).count == 30
            both = M(must=[M(key="file_path", match=M(text="/projA/")), M(key="language", match=M(value="go"))])
            want = sum(1 for i, f in enumerate(files) if "/projA/" in f and i % 3 == 0)
            assert (await s.client.count("code_chunks", count_filter=both)).count == want
            await s.client.delete("code_chunks", points_selector=M(filter=both))
            assert (await s.client.count("code_chunks", count_filter=flt)).count == 30 - want
            unc = M(must=[M(key="start_line", match=M(value=None))])               # a key the device does not code: host walk
            assert (await s.client.count("code_chunks", count_filter=unc)).count == 60 - want
            await s.client.delete("code_chunks", points_selector=M(filter=M(must=[])))          # no condition: every point
            assert (await s.get_collection_info("code_chunks")).points_count == 0
    asyncio.run(go())


def test_hits_are_resolved_inside_the_job_so_a_compaction_cannot_renumber_them(fake):
    """Round-3 advisor finding: searches returned raw SLOTS to the event loop and read ids / payloads there, while the next
    queued job (the reference indexer's delete-and-reupsert of a file, embeddings/indexer.py:61-64) could compact the tables and
    renumber every slot.  Searches, a compaction-triggering delete and the device re-rank's hit view are interleaved here: every
    hit must carry the id and payload of the row whose VECTOR was found."""
    rng = np.random.default_rng(11)
    n = 600
    vecs = rng.standard_normal((n, 768)).astype(np.float32)
    pay = [{"file_path": f"/p/f{i % 6}.py", "entity_type": "function", "entity_name": f"fn_{i}", "language": "python", "start_line": i,
            "end_line": i + 1, "content": f"body {i}", "graph_node_id": None, "content_hash": "h", "project_name": "p"} for i in range(n)]

    async def go():
        async with store_mod.HipVectorStore(dim=768, initial_capacity=1024, compact_dead_fraction=0.1, compact_min_dead=8) as st:
            await st.create_collections()
            await st.upsert("code_chunks", [f"id{i}" for i in range(n)], vecs, pay)
            keep = [i for i in range(n) if i % 6 not in (0, 1)]           # files f0 and f1 go away below: 1/3 of the rows, in front

            async def searches():
                out = []
                for i in keep[:40]:
                    out.append((i, await st.search("code_chunks", vecs[i].tolist(), limit=3)))
                    await asyncio.sleep(0)
                return out

            async def deletes():
                await asyncio.sleep(0)
                await st.delete("code_chunks", {"file_path": "/p/f0.py"})
                await st.delete("code_chunks", {"file_path": "/p/f1.py"})

            found, _ = await asyncio.gather(searches(), deletes())
            info = await st.get_collection_info("code_chunks")
            assert info.config["compactions"] >= 1 and info.points_count == len(keep)
            for i, hits in found:
                assert hits[0]["id"] == f"id{i}" and hits[0]["payload"]["entity_name"] == f"fn_{i}", (i, hits[0])
            # after the compaction: slots have moved, results have not
            for i in keep[::37]:
                top = (await st.search("code_chunks", vecs[i].tolist(), limit=1))[0]
                assert top["id"] == f"id{i}" and top["payload"]["start_line"] == i
            batch = await st.search_batch("code_chunks", vecs[keep[:5]], limit=2)
            assert [b[0]["id"] for b in batch] == [f"id{i}" for i in keep[:5]]
    asyncio.run(go())


def test_search_hits_sync_builds_hits_under_the_lock(fake):
    """The worker job itself returns hit dictionaries (never slots): what makes the test above hold by construction."""
    async def go():
        async with store_mod.HipVectorStore(dim=768, initial_capacity=64) as st:
            await st.create_collections()
            v = np.eye(768, dtype=np.float32)[:4]
            await st.upsert("code_chunks", ["a", "b", "c", "d"], v, [{"file_path": f"{c}.py"} for c in "abcd"])
            hits = await st._run(st._search_hits_sync, "code_chunks", v[:2], [1, 2], None)
            assert [len(h) for h in hits] == [1, 2] and hits[0][0]["id"] == "a" and hits[1][0]["payload"] == {"file_path": "b.py"}
    asyncio.run(go())
