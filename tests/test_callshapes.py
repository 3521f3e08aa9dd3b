"""Callers of the hot path (providers base, Embedder, VectorIndexer, both VectorSearchers) reproduce the call
shapes and outputs captured from the reference's own classes (tests/golden/callshapes_reference.json, made by
tests/golden/gen_goldens.py against AsyncMock stores/embedders).  Mirrors tests/test_embeddings.py:333-625 of the reference."""
import asyncio
import dataclasses
import json
import os
import types
from pathlib import Path
from unittest.mock import AsyncMock, MagicMock

import pytest

import coderag_amd  # noqa: F401
from coderag_amd import embedder as embedder_mod
from coderag_amd import indexer as indexer_mod
from coderag_amd import vector_search as vs_mod
from coderag_amd.errors import EmbeddingError, IndexingError, QueryError, VectorStoreError
from coderag_amd.providers import BaseEmbeddingProvider, ProviderConfig, RetryError, _backoff_seconds

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "callshapes_reference.json")))


def run(coro):
    return asyncio.run(coro)


def call_log(mock):
    return json.loads(json.dumps([[name, list(args), dict(kwargs)] for name, args, kwargs in mock.mock_calls], default=str))


def scrub(log):
    for entry in log:
        kw = entry[2]
        if "ids" in kw:
            kw["ids"] = {"n": len(kw["ids"]), "all_uuid4_str": all(isinstance(i, str) and len(i) == 36 for i in kw["ids"])}
        if "progress_callback" in kw:
            kw["progress_callback"] = None if kw["progress_callback"] in (None, "None") else "callable"
        entry[1] = ["<ParsedFile>" if isinstance(a, str) and a.startswith("<ParsedFile") else a for a in entry[1]]
    return log


class ParsedFileStub:
    """Just what VectorIndexer reads: file_info.path / .content_hash (parsing/models.py:48-58)."""

    def __init__(self):
        self.file_info = types.SimpleNamespace(path=Path("/project/main.py"), content_hash="hash123")
        self.content = "def hello(): pass"

    def __repr__(self):
        return "<ParsedFile>"


def two_chunks():
    C = indexer_mod.CodeChunk
    return [C(content="def hello(): pass", file_path="/project/main.py", entity_type="function", entity_name="hello", language="python",
              start_line=1, end_line=2),
            C(content="def world(): pass", file_path="/project/main.py", entity_type="function", entity_name="world", language="python",
              start_line=4, end_line=5, graph_node_id="world", content_hash="hash123", project_name="proj")]


def index_mocks(needs_update=True):
    q = AsyncMock()
    q.file_needs_update = AsyncMock(return_value=needs_update)
    e = AsyncMock()
    e.embed = AsyncMock(return_value=[0.1] * 4)
    e.embed_with_progress = AsyncMock(return_value=[[0.1] * 4, [0.2] * 4])
    c = MagicMock()
    c.chunk_file = MagicMock(return_value=two_chunks())
    return q, e, c


# ------------------------------------------------------------------------------------------ a-5 providers/base.py
def test_embed_batch_slicing_and_order():
    sizes = []

    class Rec(BaseEmbeddingProvider):
        async def _embed_impl(self, texts):
            sizes.append(len(texts))
            return [[float(t)] for t in texts]
    p = Rec(ProviderConfig(provider="rec", model="m"))
    out = run(p.embed_batch([str(i) for i in range(250)], batch_size=100))
    g = GOLD["a5_embed_batch"]
    assert sizes == g["impl_call_sizes"] and len(out) == g["n_out"] and [r[0] for r in out] == [float(i) for i in range(250)]
    assert run(p.embed("7")) == GOLD["a5_embed_single"]["value"] and sizes[-1] == 1


def test_embed_retries_every_exception_five_times_with_backoff():
    waits, attempts = [], []

    class Flaky(BaseEmbeddingProvider):
        async def _embed_impl(self, texts):
            attempts.append(1)
            raise EmbeddingError("deterministic failure")

    async def fake_sleep(sec):
        waits.append(sec)
    p = Flaky(ProviderConfig(provider="x", model="m"))
    p._sleep = fake_sleep
    with pytest.raises(RetryError) as e:
        run(p.embed("t"))
    assert len(attempts) == 5 and waits == [1.0, 2.0, 4.0, 8.0]       # tenacity wait_exponential(1, min=1, max=60)
    assert isinstance(e.value.last_exception, EmbeddingError)
    assert _backoff_seconds(10) == 60.0

    class Once(BaseEmbeddingProvider):
        n = 0

        async def _embed_impl(self, texts):
            Once.n += 1
            if Once.n < 3:
                raise RuntimeError("transient")
            return [[1.0]]
    p2 = Once(ProviderConfig(provider="x", model="m"))
    p2._sleep = fake_sleep
    assert run(p2.embed("t")) == [1.0] and Once.n == 3


# ------------------------------------------------------------------------------------------ a-7 embeddings/embedder.py
@pytest.mark.parametrize("n", [0, 1, 100, 101, 250])
def test_embed_with_progress(n):
    prov = MagicMock()
    prov.config = types.SimpleNamespace(provider="p", model="m")
    calls = []

    async def eb(batch, batch_size):
        calls.append([len(batch), batch_size])
        return [[0.0]] * len(batch)
    prov.embed_batch = eb
    e = embedder_mod.Embedder(provider_instance=prov)
    prov.set_concurrency.assert_called_once_with(5)
    prog = []
    out = run(e.embed_with_progress(["t"] * n, progress_callback=lambda d, t: prog.append([d, t])))
    g = GOLD["a7_embed_with_progress"][str(n)]
    assert calls == g["provider_calls"] and prog == g["progress"] and len(out) == g["n_out"]
    assert embedder_mod.OpenAIEmbedder is embedder_mod.Embedder


# ------------------------------------------------------------------------------------------ a-8 / a-9 embeddings/indexer.py
def test_chunk_payload_schema():
    assert two_chunks()[1].to_payload() == GOLD["a8_payload"]
    assert list(two_chunks()[1].to_payload()) == ["file_path", "entity_type", "entity_name", "language", "start_line", "end_line",
                                                  "content", "graph_node_id", "content_hash", "project_name"]   # chunker.py:24-37


def test_index_file_call_sequence():
    g = GOLD["a9_vector_indexer"]
    q, e, c = index_mocks()
    n = run(indexer_mod.VectorIndexer(q, e, c).index_file(ParsedFileStub(), project_name="my-project"))
    assert n == g["index_file"]["returned"] == 2
    assert scrub(call_log(q)) == g["index_file"]["store"]
    assert scrub(call_log(e)) == g["index_file"]["embedder"]
    assert scrub(call_log(c)) == g["index_file"]["chunker"]
    q.delete.assert_called_once()
    q.upsert.assert_called_once()


def test_index_file_skip_force_and_empty():
    g = GOLD["a9_vector_indexer"]
    q, e, c = index_mocks(needs_update=False)
    assert run(indexer_mod.VectorIndexer(q, e, c).index_file(ParsedFileStub())) == g["skip_unchanged"]["returned"] == 0
    assert scrub(call_log(q)) == g["skip_unchanged"]["store"] and call_log(e) == g["skip_unchanged"]["embedder"]
    q.upsert.assert_not_called()
    q, e, c = index_mocks(needs_update=False)
    assert run(indexer_mod.VectorIndexer(q, e, c).index_file(ParsedFileStub(), force=True)) == g["force"]["returned"] == 2
    assert scrub(call_log(q)) == g["force"]["store"]
    q, e, c = index_mocks()
    c.chunk_file = MagicMock(return_value=[])
    assert run(indexer_mod.VectorIndexer(q, e, c).index_file(ParsedFileStub())) == g["no_chunks"]["returned"] == 0
    assert scrub(call_log(q)) == g["no_chunks"]["store"] and call_log(e) == g["no_chunks"]["embedder"]


def test_index_files_summary_and_errors():
    g = GOLD["a9_vector_indexer"]
    q, e, c = index_mocks()
    assert run(indexer_mod.VectorIndexer(q, e, c).index_files([ParsedFileStub(), ParsedFileStub()], project_name="p")) == g["index_files"]["returned"]
    assert q.upsert.call_count == g["index_files"]["upserts"]
    q, e, c = index_mocks()
    run(indexer_mod.VectorIndexer(q, e, c).index_summary(file_path="/project/main.py", entity_type="function", entity_name="hello",
                                                         summary="This function says hello", graph_node_id="hello"))
    assert scrub(call_log(q)) == g["index_summary"]["store"] and call_log(e) == g["index_summary"]["embedder"]
    e.embed.assert_called_once_with("This function says hello")
    q, e, c = index_mocks()
    e.embed_with_progress.side_effect = Exception("API Error")
    with pytest.raises(IndexingError) as ex:
        run(indexer_mod.VectorIndexer(q, e, c).index_file(ParsedFileStub()))
    assert {"type": "IndexingError", "stage": ex.value.stage, "str": str(ex.value)} == g["error"]
    q, e, c = index_mocks()
    e.embed_with_progress.side_effect = Exception("API Error")
    assert run(indexer_mod.VectorIndexer(q, e, c).index_files([ParsedFileStub(), ParsedFileStub()])) == g["index_files_swallows"]["returned"] == 0


# ------------------------------------------------------------------------------------------ a-11 / a-12 searchers
HITS = [{"id": "1", "score": 0.95, "payload": {"file_path": "/project/main.py", "entity_type": "function", "entity_name": "hello",
                                                "content": "def hello(): pass", "start_line": 1, "end_line": 2, "language": "python",
                                                "graph_node_id": "hello"}},
        {"id": "2", "score": 0.85, "payload": {"file_path": "/project/utils.py", "entity_name": "helper"}},
        {"id": "3", "score": 0.80, "payload": {"file_path": "a.py", "entity_name": "in_a", "summary": "sum"}}]


def search_mocks():
    q = AsyncMock()
    q.search = AsyncMock(return_value=HITS)
    e = AsyncMock()
    e.embed = AsyncMock(return_value=[0.5, 0.5])
    return q, e


def test_indexer_flavour_searcher():
    g = GOLD["a11_indexer_searcher"]
    q, e = search_mocks()
    r = run(indexer_mod.VectorSearcher(q, e).search_code("hello world", limit=7))
    assert call_log(q) == g["plain"]["store"] and call_log(e) == g["plain"]["embedder"]
    assert [dataclasses.asdict(x) for x in r] == g["plain"]["results"]
    assert isinstance(r[0], indexer_mod.CodeSearchResult) and r[0].score == 0.95 and r[0].entity_name == "hello"
    q, e = search_mocks()
    run(indexer_mod.VectorSearcher(q, e).search_code("hello", language="python", entity_type="function", project_name="my-project"))
    assert call_log(q) == g["filters"]["store"]
    q, e = search_mocks()
    r = run(indexer_mod.VectorSearcher(q, e).search_summaries("greeting", entity_type="class"))
    assert call_log(q) == g["summaries"]["store"] and [dataclasses.asdict(x) for x in r] == g["summaries"]["results"]
    q, e = search_mocks()
    e.embed.side_effect = Exception("API Error")
    with pytest.raises(IndexingError) as ex:
        run(indexer_mod.VectorSearcher(q, e).search_code("test query"))
    assert {"stage": ex.value.stage, "str": str(ex.value)} == g["error"]


def test_query_flavour_searcher():
    g = GOLD["a12_query_searcher"]
    q, e = search_mocks()
    r = run(vs_mod.VectorSearcher(q, e).search_code("find auth", limit=4))
    assert call_log(q) == g["plain"]["store"] and call_log(e) == g["plain"]["embedder"] and r == g["plain"]["results"]
    assert list(r[0]) == ["score", "file_path", "entity_type", "entity_name", "language", "content", "start_line", "end_line",
                          "graph_node_id"]                                   # vector_search.py:230-241
    q, e = search_mocks()
    run(vs_mod.VectorSearcher(q, e).search_code("find auth", language="python"))
    assert call_log(q) == g["language"]["store"]
    q, e = search_mocks()
    r = run(vs_mod.VectorSearcher(q, e).search_summaries("what", limit=3, project_name="proj"))
    assert call_log(q) == g["summaries"]["store"] and r == g["summaries"]["results"]
    q, e = search_mocks()
    r = run(vs_mod.VectorSearcher(q, e).find_similar_code("def f(): pass", limit=1, exclude_file="a.py"))
    assert call_log(q) == g["similar_exclude"]["store"] and r == g["similar_exclude"]["results"]
    assert q.search.call_args.kwargs["limit"] == 1 + vs_mod.EXCLUDE_FILE_BUFFER
    q, e = search_mocks()
    r = run(vs_mod.VectorSearcher(q, e).find_similar_code("def f(): pass", limit=2))
    assert call_log(q) == g["similar_plain"]["store"] and r == g["similar_plain"]["results"]


def test_query_flavour_error_mapping():
    g = GOLD["a12_query_searcher"]["errors"]
    got = {}
    for label, call in (("blank_code", lambda s: s.search_code("   ")), ("blank_summary", lambda s: s.search_summaries("")),
                        ("blank_similar", lambda s: s.find_similar_code("\n"))):
        q, e = search_mocks()
        with pytest.raises(QueryError) as ex:
            run(call(vs_mod.VectorSearcher(q, e)))
        got[label] = str(ex.value)
        e.embed.assert_not_called()
    for label, exc, call in (("embed_code", EmbeddingError("boom"), lambda s: s.search_code("x")),
                             ("store_code", VectorStoreError("down"), lambda s: s.search_code("x")),
                             ("store_summary", VectorStoreError("down"), lambda s: s.search_summaries("x")),
                             ("embed_similar", EmbeddingError("boom"), lambda s: s.find_similar_code("x")),
                             ("store_similar", VectorStoreError("down"), lambda s: s.find_similar_code("x"))):
        q, e = search_mocks()
        if label.startswith("embed"):
            e.embed.side_effect = exc
        else:
            q.search.side_effect = exc
        with pytest.raises(QueryError) as ex:
            run(call(vs_mod.VectorSearcher(q, e)))
        got[label] = str(ex.value)
    assert got == g
    q, e = search_mocks()
    e.embed.side_effect = ValueError("not mapped")                            # other exceptions pass through unchanged
    with pytest.raises(ValueError):
        run(vs_mod.VectorSearcher(q, e).search_code("x"))
