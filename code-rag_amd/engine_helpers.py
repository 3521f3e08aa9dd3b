"""The two behaviours of ``QueryEngine`` that shape what reaches the ranker, restated as free functions
(``src/lattice/query/engine.py:315-377``).  The engine itself (LLM planning, Memgraph, response writing) is
out of scope; these helpers let a batch harness feed the ranker exactly what the reference would."""

from __future__ import annotations

import asyncio
from typing import Any, Awaitable, Callable

from .query_types import QueryIntent, intent_key
from .settings import get_settings

_SUMMARY_INTENTS = {i.value for i in (QueryIntent.EXPLAIN_IMPLEMENTATION, QueryIntent.EXPLAIN_RELATIONSHIP,
                                      QueryIntent.EXPLAIN_DATA_FLOW, QueryIntent.EXPLAIN_ARCHITECTURE,
                                      QueryIntent.SEARCH_FUNCTIONALITY)}


async def execute_vector_search(vector_searcher, query: str, plan, limit: int, language: str | None,
                                project_name: str | None = None) -> list[dict[str, Any]]:
    """engine.py:315-346: code hits capped at ``max_vector_results``; explanatory/functional intents also get
    ``limit // 2`` summary hits appended."""
    cap = get_settings().max_vector_results
    results = await vector_searcher.search_code(query=query, limit=min(limit, cap), language=language, project_name=project_name)
    if intent_key(plan.primary_intent) in _SUMMARY_INTENTS:
        results.extend(await vector_searcher.search_summaries(query=query, limit=limit // 2, project_name=project_name))
    return results


def centrality_candidates(graph_context, vector_results: list[dict[str, Any]]) -> list[str]:
    """engine.py:353-363: names of <= 5 primary graph entities and of the first 5 vector hits (graph node id, else
    entity name), de-duplicated, truncated to ``max_centrality_lookups``.  (The reference iterates a ``set``, so
    which names survive the truncation is hash-order dependent there; insertion order is used here.)"""
    names: dict[str, None] = {}
    for node in graph_context.primary_entities[:5]:
        names[node.qualified_name or node.name] = None
    for hit in vector_results[:5]:
        if hit.get("entity_name"):
            names[hit.get("graph_node_id") or hit.get("entity_name")] = None
    return list(names)[: get_settings().max_centrality_lookups]


async def get_centrality_scores(lookup: Callable[[str], Awaitable[dict[str, int]]], graph_context,
                                vector_results: list[dict[str, Any]]) -> dict[str, dict[str, int]]:
    """engine.py:348-377: one degree lookup per candidate, failures dropped."""
    names = centrality_candidates(graph_context, vector_results)
    if not names:
        return {}
    answers = await asyncio.gather(*(lookup(n) for n in names), return_exceptions=True)
    return {n: a for n, a in zip(names, answers) if isinstance(a, dict)}


async def search_and_rank_batch(vector_searcher, ranker, queries, plans, graph_contexts=None, centrality=None,
                                limit: int = 20, language: str | None = None, project_name: str | None = None):
    """BASELINE config 5 in one call: ONE batched corpus scan for all queries (``search_code_batch``), then the hybrid
    re-rank of each query's merged top-k exactly as ``QueryEngine.search`` would do it one query at a time
    (engine.py:222-260: vector limit capped at ``max_vector_results``, then ``HybridRanker.rank_results``).
    ``queries`` are strings or ready vectors; ``plans[i]`` / ``graph_contexts[i]`` / ``centrality[i]`` belong to query i."""
    from .query_types import GraphContext
    cap = get_settings().max_vector_results
    per_query = await vector_searcher.search_code_batch(queries, limit=min(limit, cap), language=language, project_name=project_name)
    ranked = []
    for i, hits in enumerate(per_query):
        ctx = graph_contexts[i] if graph_contexts is not None and graph_contexts[i] is not None else GraphContext()
        ranked.append(ranker.rank_results(plans[i], ctx, hits, (centrality[i] if centrality is not None else None)))
    return ranked


async def search_and_rank_batch_device(store, reranker, host_ranker, query_vectors, plans, limit: int = 20,
                                       language: str | None = None, project_name: str | None = None):
    """:func:`search_and_rank_batch` for vector-only queries with the re-rank on the device: ONE corpus scan, one
    ``crh_rerank_vector`` launch for the whole batch, and payload dictionaries touched only for the <= 50 survivors of each
    query.  A query the device declines (``count == -1``: more than 8 plan entities, an over-long name) goes through
    ``host_ranker`` on its full hit list, so results equal :func:`search_and_rank_batch` in every case."""
    from .query_types import GraphContext
    from .ranking.device import DeviceReranker
    from .store import CollectionName
    from .vector_search import _CODE_KEYS, _project
    cap = get_settings().max_vector_results
    filters = {k: v for k, v in (("language", language), ("project_name", project_name)) if v}
    col, out, rows, scores = await store.search_rerank_batch(CollectionName.CODE_CHUNKS.value, query_vectors, plans, reranker,
                                                             limit=min(limit, cap), filters=filters or None)

    class _Hits:   # candidate position -> the flattened hit dict of query/vector_search.py:111-131, built on demand
        def __init__(self, q):
            self.q = q

        def __getitem__(self, i):
            return _project(col.hit(int(rows[self.q, i]), float(scores[self.q, i])), _CODE_KEYS)

    ranked = []
    for q in range(len(plans)):
        if out.count[q] >= 0:
            ranked.append(DeviceReranker.materialise(out, q, _Hits(q)))
            continue
        hits = [_project(col.hit(int(r), float(s)), _CODE_KEYS) for r, s in zip(rows[q], scores[q]) if r >= 0]
        names = centrality_candidates(GraphContext(), hits)
        deg = col._degrees or {}
        ranked.append(host_ranker.rank_results(plans[q], GraphContext(), hits, {n: {"total_degree": deg[n]} for n in names if n in deg}))
    return ranked
