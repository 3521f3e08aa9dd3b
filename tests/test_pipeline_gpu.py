"""End-to-end on the GPU through the reference-shaped surfaces only: chunk -> embed (HIP encoder behind the provider)
-> upsert (HIP index behind the store) -> semantic search -> hybrid re-rank -> MCP tool; then incremental re-index,
snapshot/restore.  Every number a caller sees must be reproducible from the pieces: the vectors the store returns scores
for are the provider's vectors, and the scores equal the oracle's cosine on them."""
import asyncio
import types
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _parsed_file(path, funcs, content_hash="h1"):
    ents = [types.SimpleNamespace(type=types.SimpleNamespace(value="function"), name=n, qualified_name=f"{Path(path).stem}.{n}",
                                  signature=f"def {n}(x)", docstring=f"Compute {n}.", code=code, start_line=10 * i + 1,
                                  end_line=10 * i + 8) for i, (n, code) in enumerate(funcs)]
    info = types.SimpleNamespace(path=Path(path), content_hash=content_hash, language=types.SimpleNamespace(value="python"))
    return types.SimpleNamespace(file_info=info, content="\n".join(c for _, c in funcs), all_entities=ents)


def test_index_search_rank_roundtrip(gpu, tmp_path):
    import coderag_amd  # noqa: F401
    from coderag_amd.embedder import Embedder
    from coderag_amd.indexer import CodeChunker, VectorIndexer
    from coderag_amd.mcp_tools import create_semantic_search_tool
    from coderag_amd.providers import HipUniXcoderProvider, ProviderConfig
    from coderag_amd.query_types import ExtractedEntity, GraphContext, QueryIntent, QueryPlan
    from coderag_amd.ranking import HybridRanker
    from coderag_amd.store import HipVectorStore
    from coderag_amd.vector_search import VectorSearcher
    from oracle import search as orc

    provider = HipUniXcoderProvider(ProviderConfig(provider="unixcoder-hip", model="synthetic",
                                                   extra={"synthetic_weights": 5, "num_layers": 2}))
    embedder = Embedder(provider_instance=provider)
    files = [_parsed_file(f"/proj/mod{f}.py", [(f"fn_{f}_{i}", f"def fn_{f}_{i}(x):\n    y = x * {i} + {f}\n    return helper_{i % 3}(y)\n" * (1 + i % 4))
                                                for i in range(6)]) for f in range(5)]

    async def go():
        async with HipVectorStore(dim=embedder.embedding_dim, dtype="f32", initial_capacity=64) as store:
            await store.create_collections()
            indexer = VectorIndexer(store, embedder, CodeChunker(max_tokens=1000, overlap_tokens=200))
            progress = []
            n = await indexer.index_files(files, progress_callback=lambda d, t: progress.append((d, t)), project_name="proj")
            assert n == 30 and progress[-1] == (5, 5)
            assert (await store.get_collection_info("code_chunks")).points_count == 30
            assert await indexer.index_file(files[0], project_name="proj") == 0          # unchanged hash: skipped
            assert await indexer.index_file(files[0], force=True, project_name="proj") == 6
            assert (await store.get_collection_info("code_chunks")).points_count == 30    # old chunks of the file were deleted first

            searcher = VectorSearcher(store, embedder)
            query = files[2].all_entities[3].code                                          # a stored chunk's own code
            hits = await searcher.search_code(query, limit=5, language="python", project_name="proj")
            assert len(hits) == 5 and set(hits[0]) == {"score", "file_path", "entity_type", "entity_name", "language", "content",
                                                         "start_line", "end_line", "graph_node_id"}
            assert [h["score"] for h in hits] == sorted((h["score"] for h in hits), reverse=True)
            assert hits[0]["file_path"] == "/proj/mod2.py" and query in hits[0]["content"]
            # the scores are the oracle's cosine between the provider's own vectors
            qv = np.asarray(await embedder.embed(query), np.float32)
            cv = np.asarray(await embedder.embed_batch([h["content"] for h in hits]), np.float32)
            es, _ = orc.cosine_search(cv, qv[None], 5)
            assert np.allclose(sorted((h["score"] for h in hits), reverse=True), es[0], atol=2e-6)
            similar = await searcher.find_similar_code(query, limit=3, exclude_file="/proj/mod2.py")
            assert len(similar) == 3 and all(s["file_path"] != "/proj/mod2.py" for s in similar)

            plan = QueryPlan("how does fn_2_3 work", QueryIntent.EXPLAIN_IMPLEMENTATION, entities=[ExtractedEntity("fn_2_3")])
            ranked = HybridRanker().rank_results(plan, GraphContext(), hits, {"mod2.fn_2_3": {"total_degree": 25}})
            assert ranked[0].entity_name == "mod2.fn_2_3" and ranked[0].source == "vector"
            assert ranked[0].signal_scores["query_entity_match"] == 0.5 and ranked[0].signal_scores["centrality"] == 0.5

            tool = create_semantic_search_tool(lambda: searcher)
            res = await tool["function"]("multiply and call helper", limit=4, entity_type="function")
            assert res.success and len(res.data) == 4 and set(res.data[0]) == {"qualified_name", "entity_type", "file_path", "score", "summary"}

            def stored_text(e):                      # what the chunker stores for an entity (chunker.py:128-135)
                return "\n".join([e.signature, f'"""{e.docstring}"""', e.code])
            batch = await searcher.search_code_batch([stored_text(f.all_entities[0]) for f in files], limit=3)
            assert [b[0]["file_path"] for b in batch] == [str(f.file_info.path) for f in files]
            assert all(abs(b[0]["score"] - 1.0) < 1e-5 for b in batch)       # a stored text finds itself

            await store.save(str(tmp_path / "snap"))
            await store.load(str(tmp_path / "snap"))
            again = await searcher.search_code(query, limit=5, language="python", project_name="proj")
            assert again == hits
    asyncio.run(go())
    assert provider.submissions >= 1


def test_reindexing_the_same_files_five_times_does_not_grow_the_index(gpu):
    """The reference's indexing flow deletes and re-inserts every chunk of a file on every run (embeddings/indexer.py:61-64,
    called with force=True from pipeline/orchestrator.py:630-650).  200 files re-indexed five times through
    VectorIndexer.index_file(force=True): the store compacts by itself, the scan reads only live rows afterwards
    (crh_search_stats.rows == alive rows), and search results are bit-identical to a store built once from the same files."""
    import coderag_amd  # noqa: F401
    from coderag_amd.embedder import Embedder
    from coderag_amd.indexer import CodeChunker, VectorIndexer
    from coderag_amd.providers import HipUniXcoderProvider, ProviderConfig
    from coderag_amd.store import HipVectorStore
    from coderag_amd.vector_search import VectorSearcher

    provider = HipUniXcoderProvider(ProviderConfig(provider="unixcoder-hip", model="synthetic", extra={"synthetic_weights": 5, "num_layers": 2}))
    embedder = Embedder(provider_instance=provider)
    files = [_parsed_file(f"/proj/pkg{f % 7}/mod{f}.py", [(f"fn_{f}_{i}", f"def fn_{f}_{i}(x):\n    y = x * {i} + {f}\n    return helper_{i % 3}(y)\n" * (1 + i % 3))
                                                          for i in range(4)]) for f in range(200)]
    queries = [files[f].all_entities[f % 4].code for f in (3, 77, 150, 199)]

    async def build(rounds, **kw):
        async with HipVectorStore(dim=embedder.embedding_dim, dtype="f32", initial_capacity=256, **kw) as store:
            await store.create_collections()
            indexer = VectorIndexer(store, embedder, CodeChunker(max_tokens=1000, overlap_tokens=200))
            for _ in range(rounds):
                for f in files:
                    assert await indexer.index_file(f, force=True, project_name="proj") == 4
            info = await store.get_collection_info("code_chunks")
            searcher = VectorSearcher(store, embedder)
            hits = [await searcher.search_code(q, limit=10) for q in queries]
            col = store._col("code_chunks")
            return info, hits, col.shards.stats()["rows"], col.payloads.n, col.ids.n

    info5, hits5, scanned5, pn5, in5 = asyncio.run(build(5, compact_dead_fraction=0.25, compact_min_dead=100))
    info1, hits1, scanned1, _, _ = asyncio.run(build(1))
    assert info5.points_count == info1.points_count == 800
    assert info5.config["compactions"] >= 2 and info5.config["rows_appended"] < 2 * 800         # dead rows were reclaimed on the way
    assert pn5 == in5 == info5.config["rows_appended"]                                         # ... in the host tables too
    assert hits5 == hits1                                                                      # same ids' payloads, same score bits
    # an explicit compaction leaves exactly the alive rows for the scan to stream
    async def compacted():
        async with HipVectorStore(dim=embedder.embedding_dim, dtype="f32", initial_capacity=256, compact_dead_fraction=0.0) as store:
            await store.create_collections()
            indexer = VectorIndexer(store, embedder, CodeChunker(max_tokens=1000, overlap_tokens=200))
            for _ in range(3):
                for f in files[:50]:
                    await indexer.index_file(f, force=True, project_name="proj")
            assert (await store.get_collection_info("code_chunks")).config["rows_appended"] == 600     # never compacted: 3 x 200 rows
            assert await store.compact("code_chunks") == 400
            searcher = VectorSearcher(store, embedder)
            await searcher.search_code(queries[0], limit=5)
            st = store._col("code_chunks").shards.stats()
            assert st["rows"] == 200 == (await store.get_collection_info("code_chunks")).points_count
            import os, tempfile
            with tempfile.TemporaryDirectory() as d:
                await store.save(d)
                n = np.fromfile(os.path.join(d, "code_chunks", "alive.u32"), np.uint32)
                assert int(sum(bin(int(w)).count("1") for w in n)) == 200 and len(n) == 7       # 200 rows = 7 tiles, every stored row alive
    asyncio.run(compacted())


def test_batched_indexing_and_lazy_upsert_on_the_gpu(gpu):
    """`VectorIndexer.index_files_batched` (one pass: chunk -> ONE coalesced embedding submission as an array -> one upsert) and
    the store-side form (`embed_in_store=True`: `upsert(vectors=None, texts=..., embed=provider.embed_texts_sync)`, here on three
    in-process shards) leave the store exactly as the reference's per-file flow does: same chunks and payloads, the same hits
    with the same f32 scores for a query (embeddings do not depend on the batch they travelled in)."""
    import coderag_amd  # noqa: F401
    from coderag_amd.embedder import Embedder
    from coderag_amd.indexer import CodeChunker, VectorIndexer
    from coderag_amd.providers import HipUniXcoderProvider, ProviderConfig
    from coderag_amd.store import HipVectorStore
    from coderag_amd.vector_search import VectorSearcher
    provider = HipUniXcoderProvider(ProviderConfig(provider="unixcoder-hip", model="synthetic", extra={"synthetic_weights": 5, "num_layers": 2}))
    embedder = Embedder(provider_instance=provider)
    files = [_parsed_file(f"/proj/mod{f}.py", [(f"fn_{f}_{i}", f"def fn_{f}_{i}(x):\n    y = x * {i} + {f}\n    return helper_{i % 3}(y)\n" * (1 + i % 4))
                                                for i in range(5 + f % 3)]) for f in range(9)]
    query = files[4].all_entities[2].code

    async def run(mode, shards):
        async with HipVectorStore(dim=embedder.embedding_dim, dtype="f32", initial_capacity=64, shards=shards) as store:
            await store.create_collections()
            indexer = VectorIndexer(store, embedder, CodeChunker(max_tokens=1000, overlap_tokens=200))
            if mode == "sequential":
                n = await indexer.index_files(files, project_name="proj")
            else:
                n = await indexer.index_files_batched(files, project_name="proj", embed_in_store=(mode == "in_store"))
            assert await indexer.index_files_batched(files, project_name="proj") == 0                     # unchanged: skipped
            hits = await VectorSearcher(store, embedder).search_code(query, limit=8, project_name="proj")
            return n, [(h["entity_name"], h["file_path"], h["start_line"], h["content"], np.float32(h["score"]).tobytes()) for h in hits]
    seq = asyncio.run(run("sequential", 1))
    bat = asyncio.run(run("batched", 1))
    ins = asyncio.run(run("in_store", 3))
    assert seq[0] == bat[0] == ins[0] == sum(5 + f % 3 for f in range(9))
    assert seq[1] == bat[1] == ins[1] and seq[1][0][0] == "mod4.fn_4_2"
