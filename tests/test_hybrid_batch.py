"""Config-5 path on the CPU tier: batched vector search + hybrid re-rank equals the one-query-at-a-time flow of the
reference (QueryEngine.search: vector search -> HybridRanker), and the MCP tool / engine helpers behave as restated."""
import asyncio

import numpy as np
import pytest

import coderag_amd  # noqa: F401
from coderag_amd import ffi
from coderag_amd import engine_helpers as eh
from coderag_amd.mcp_tools import create_semantic_search_tool
from coderag_amd.query_types import ExtractedEntity, GraphContext, GraphNode, QueryIntent, QueryPlan
from coderag_amd.ranking import HybridRanker
from coderag_amd.store import HipVectorStore
from coderag_amd.vector_search import VectorSearcher
from tests.fake_index import FakeIndex


class VecEmbedder:
    """Deterministic text -> vector stand-in (the encoder has its own tests)."""

    def _v(self, t):
        return np.random.default_rng(abs(hash(t)) % (2 ** 32)).standard_normal(768).astype(np.float32)

    async def embed(self, text):
        return self._v(text).tolist()

    async def embed_batch(self, texts, batch_size=100):
        return [self._v(t).tolist() for t in texts]


@pytest.fixture
def store(monkeypatch):
    monkeypatch.setattr(ffi, "Index", FakeIndex)
    monkeypatch.setattr(ffi, "lib", lambda: object())
    monkeypatch.setattr(ffi, "device_count", lambda: 1)
    monkeypatch.setattr(ffi, "device_info", lambda d=0: {"name": "fake", "arch": "gfx950", "hbm_bytes": 0, "cu_count": 256})
    return HipVectorStore(dim=768, initial_capacity=256)


def test_batched_search_and_rank_equals_sequential(store):
    async def go():
        rng = np.random.default_rng(1)
        n = 300
        async with store as s:
            await s.create_collections()
            payloads = [{"file_path": f"src/f{i % 7}.py", "entity_type": "function", "entity_name": f"fn_{i}", "language": "python",
                         "start_line": i, "end_line": i + 4, "content": "x" * (40 + 13 * (i % 50)), "graph_node_id": f"mod.fn_{i}",
                         "content_hash": "h", "project_name": "p"} for i in range(n)]
            await s.upsert("code_chunks", [f"id{i}" for i in range(n)], rng.standard_normal((n, 768)).astype(np.float32).tolist(), payloads)
            searcher, ranker = VectorSearcher(s, VecEmbedder()), HybridRanker()
            queries = ["how does fn_3 work", "find code similar to fn_10", "where is fn_200", "what calls fn_7"]
            intents = [QueryIntent.EXPLAIN_IMPLEMENTATION, QueryIntent.FIND_SIMILAR, QueryIntent.LOCATE_ENTITY, QueryIntent.FIND_CALLERS]
            plans = [QueryPlan(q, it, entities=[ExtractedEntity(q.split()[-2] if q.endswith("work") else q.split()[-1])]) for q, it in zip(queries, intents)]
            ctxs = [GraphContext(primary_entities=[GraphNode("function", "fn_3", "mod.fn_3", "src/f3.py", start_line=3)]), None, None,
                    GraphContext(callers=[GraphNode("function", "fn_9", "mod.fn_9", "src/f2.py", start_line=9, metadata={"depth": 2})])]
            cent = [{"mod.fn_3": {"total_degree": 40}}, None, {}, None]
            batched = await eh.search_and_rank_batch(searcher, ranker, queries, plans, ctxs, cent, limit=30, language="python")
            for i, q in enumerate(queries):
                hits = await searcher.search_code(q, limit=min(30, 20), language="python")          # engine.py:327 cap
                one = ranker.rank_results(plans[i], ctxs[i] or GraphContext(), hits, cent[i])
                got = [(r.get_key(), r.final_score, r.source, r.signal_scores) for r in batched[i]]
                assert got == [(r.get_key(), r.final_score, r.source, r.signal_scores) for r in one]
            # ready vectors instead of strings give the same hits
            vecs = np.asarray(await VecEmbedder().embed_batch(queries), dtype=np.float32)
            assert await searcher.search_code_batch(vecs, limit=5) == await searcher.search_code_batch(queries, limit=5)
    asyncio.run(go())


def test_engine_helpers_restate_query_engine(store):
    calls = []

    class Searcher:
        async def search_code(self, **kw):
            calls.append(("code", kw))
            return [{"entity_name": f"e{i}", "graph_node_id": f"g{i}" if i % 2 else None} for i in range(7)]

        async def search_summaries(self, **kw):
            calls.append(("summaries", kw))
            return [{"entity_name": "s"}]

    async def go():
        plan = QueryPlan("q", QueryIntent.EXPLAIN_ARCHITECTURE)
        res = await eh.execute_vector_search(Searcher(), "q", plan, limit=50, language=None, project_name="proj")
        assert calls[0] == ("code", {"query": "q", "limit": 20, "language": None, "project_name": "proj"})     # min(limit, max_vector_results)
        assert calls[1] == ("summaries", {"query": "q", "limit": 25, "project_name": "proj"}) and len(res) == 8  # limit // 2, appended
        calls.clear()
        await eh.execute_vector_search(Searcher(), "q", QueryPlan("q", QueryIntent.FIND_CALLERS), 10, "python")
        assert [c[0] for c in calls] == ["code"] and calls[0][1]["limit"] == 10
        ctx = GraphContext(primary_entities=[GraphNode("class", f"C{i}", f"m.C{i}" if i else "", "f.py") for i in range(8)])
        names = eh.centrality_candidates(ctx, res)
        assert names == ["C0", "m.C1", "m.C2", "m.C3", "m.C4", "e0", "g1", "e2", "g3", "e4"]                     # 5 + 5, capped at 10

        async def lookup(name):
            if name == "m.C2":
                raise RuntimeError("graph down")
            return {"in_degree": 1, "out_degree": 2, "total_degree": 3, "relationship_count": 3}
        scores = await eh.get_centrality_scores(lookup, ctx, res)
        assert "m.C2" not in scores and len(scores) == 9 and scores["e0"]["total_degree"] == 3
        assert await eh.get_centrality_scores(lookup, GraphContext(), []) == {}
    asyncio.run(go())


def test_mcp_semantic_search_tool(store):
    class Searcher:
        async def search_code(self, query, limit, entity_type):
            if query == "boom":
                raise ValueError("no index")
            return [{"score": 0.9, "file_path": "a.py", "entity_type": entity_type or "function", "entity_name": "mod.f", "content": "..."}][:limit]

    tool = create_semantic_search_tool(lambda: Searcher())
    assert tool["name"] == "semantic_search" and tool["parameters"]["query"]["required"] is True

    async def go():
        ok = await tool["function"]("error handling", limit=5, entity_type="method")
        assert ok.success and ok.data == [{"qualified_name": "mod.f", "entity_type": "method", "file_path": "a.py", "score": 0.9, "summary": None}]
        assert ok.message == "Found 1 matches for 'error handling'."
        bad = await tool["function"]("boom")
        assert bad.success is False and bad.error == "no index" and bad.data is None
    asyncio.run(go())


def test_side_columns_host_coding():
    """The per-row side data of the device re-rank, as coded on the host (no GPU needed until it is gathered)."""
    import numpy as np
    from coderag_amd.ranking.device import SideColumns, merge_key, node_key
    payloads = [
        {"file_path": "a.py", "entity_name": "Repo.Save", "start_line": 3, "content": "x" * 120, "graph_node_id": "pkg.Repo.Save"},
        {"file_path": "a.py", "entity_name": "Repo.Save", "start_line": 3, "content": None},                       # same merge key, no node id
        {"file_path": "b.py", "entity_name": "", "start_line": None, "content": ""},
        None,                                                                                                       # tombstoned row
        {"file_path": "b.py", "entity_name": "Löwe" + "x" * 70, "start_line": 1, "content": "y", "graph_node_id": "pkg.Repo.Save"},
    ]
    side = SideColumns(0)
    side.append(payloads[:2])
    side.append(payloads[2:])
    h = {c: v[: side.rows].tolist() for c, v in side._host.items()}          # (the host arrays carry headroom past `rows`)
    assert side.rows == 5 and h["content_len"] == [120, 0, 0, 0, 1]
    assert h["key_code"][0] == h["key_code"][1] != h["key_code"][2]
    assert h["file_code"][0] == h["file_code"][1] and h["file_code"][2] == h["file_code"][4]
    assert h["node_code"][0] == h["node_code"][4] != h["node_code"][1]            # graph_node_id wins over entity_name
    assert h["name_len"] == [9, 9, 0, 0, len(("löwe" + "x" * 70).encode())]
    assert side._names[0].tobytes().rstrip(b"\x00") == b"repo.save" and side._names.shape[1] == 64                  # lower-cased, cut to 64 bytes
    assert side._names[4].tobytes() == ("löwe" + "x" * 70).encode()[:64]
    assert h["degree"] == [-1] * 5
    side.set_degrees({"pkg.Repo.Save": 12, "Repo.Save": 3})
    assert side._host["degree"][:5].tolist() == [12, 3, -1, -1, 12]
    assert node_key(payloads[1]) == "Repo.Save" and merge_key(payloads[2]) == "b.py::None"
