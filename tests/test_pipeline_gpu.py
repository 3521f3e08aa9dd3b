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
