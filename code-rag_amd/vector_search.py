"""Dict-returning semantic searcher -- the query-side surface of ``src/lattice/query/vector_search.py:43-280``.

Single-query methods keep the reference's signatures, result keys, validation and error mapping
(``EmbeddingError`` / ``VectorStoreError`` -> ``QueryError``; blank input -> ``QueryError``).  Added here: batch
entry points (``search_code_batch``) that put up to 64 queries through ONE corpus scan, which is what the
MI355X kernels are built for; the reference issues one RPC per query.
"""

from __future__ import annotations

import logging
from dataclasses import dataclass
from typing import Any

import numpy as np

from .errors import EmbeddingError, QueryError, VectorStoreError
from .store import CollectionName

logger = logging.getLogger(__name__)

DEFAULT_SEARCH_LIMIT = 10
EXCLUDE_FILE_BUFFER = 5

_CODE_KEYS = ("file_path", "entity_type", "entity_name", "language", "content", "start_line", "end_line", "graph_node_id")
_SUMMARY_KEYS = ("file_path", "entity_type", "entity_name", "summary", "graph_node_id")
_SIMILAR_KEYS = ("file_path", "entity_type", "entity_name", "content", "start_line", "end_line")


@dataclass
class CodeSearchResult:
    score: float
    file_path: str
    entity_type: str
    entity_name: str
    content: str
    language: str | None = None
    start_line: int | None = None
    end_line: int | None = None
    graph_node_id: str | None = None


@dataclass
class SummarySearchResult:
    score: float
    file_path: str
    entity_type: str
    entity_name: str
    summary: str
    graph_node_id: str | None = None


_NO_FILTER_KWARG = object()


def _project(hit: dict, keys: tuple[str, ...]) -> dict:
    payload = hit["payload"]
    row = {"score": hit["score"]}
    for k in keys:
        row[k] = payload.get(k)
    return row


class VectorSearcher:
    def __init__(self, qdrant, embedder):
        self.qdrant = qdrant
        self.embedder = embedder

    async def _lookup(self, text: str, collection: str, limit: int, filters: Any, embed_fail: str, store_fail: str):
        """Embed, search, map the two error kinds.  ``filters=_NO_FILTER_KWARG`` omits the keyword altogether, as
        the reference's ``find_similar_code`` does (vector_search.py:193-197)."""
        try:
            vector = await self.embedder.embed(text)
            kwargs = {} if filters is _NO_FILTER_KWARG else {"filters": filters}
            return await self.qdrant.search(collection=collection, query_vector=vector, limit=limit, **kwargs)
        except EmbeddingError as e:
            logger.error(f"Embedding error: {e}")
            raise QueryError(embed_fail, cause=e)
        except VectorStoreError as e:
            logger.error(f"Vector store error: {e}")
            raise QueryError(store_fail, cause=e)

    async def search_code(self, query: str, limit: int = DEFAULT_SEARCH_LIMIT, language: str | None = None,
                          entity_type: str | None = None, project_name: str | None = None) -> list[dict]:
        """vector_search.py:60-116."""
        if not query or not query.strip():
            raise QueryError("Search query cannot be empty")
        filters = {k: v for k, v in (("language", language), ("entity_type", entity_type), ("project_name", project_name)) if v}
        hits = await self._lookup(query, CollectionName.CODE_CHUNKS.value, limit, filters or None,
                                  "Failed to embed search query", "Failed to search code")
        return [_project(h, _CODE_KEYS) for h in hits]

    async def search_summaries(self, query: str, limit: int = DEFAULT_SEARCH_LIMIT, project_name: str | None = None) -> list[dict]:
        """vector_search.py:118-166 (filters on ``project_name``, which summary payloads never carry: quirk Q6)."""
        if not query or not query.strip():
            raise QueryError("Search query cannot be empty")
        filters = {"project_name": project_name} if project_name else None
        hits = await self._lookup(query, CollectionName.SUMMARIES.value, limit, filters,
                                  "Failed to embed search query", "Failed to search summaries")
        return [_project(h, _SUMMARY_KEYS) for h in hits]

    async def find_similar_code(self, code_snippet: str, limit: int = DEFAULT_SEARCH_LIMIT, exclude_file: str | None = None) -> list[dict]:
        """vector_search.py:168-219: over-fetch by 5 when a file is excluded, drop its chunks, keep ``limit``."""
        if not code_snippet or not code_snippet.strip():
            raise QueryError("Code snippet cannot be empty")
        fetch = limit + EXCLUDE_FILE_BUFFER if exclude_file else limit
        hits = await self._lookup(code_snippet, CollectionName.CODE_CHUNKS.value, fetch, _NO_FILTER_KWARG,
                                  "Failed to embed code snippet", "Failed to find similar code")
        kept = []
        for h in hits:
            if exclude_file and h["payload"].get("file_path") == exclude_file:
                continue
            kept.append(_project(h, _SIMILAR_KEYS))
            if len(kept) >= limit:
                break
        return kept

    # ------------------------------------------------------------------ batch entry (not in the reference)
    async def search_code_batch(self, queries, limit: int = DEFAULT_SEARCH_LIMIT, language: str | None = None,
                                entity_type: str | None = None, project_name: str | None = None) -> list[list[dict]]:
        """``queries``: list of strings (embedded in one provider batch) or an array [B, dim] of ready vectors."""
        filters = {k: v for k, v in (("language", language), ("entity_type", entity_type), ("project_name", project_name)) if v}
        try:
            if isinstance(queries, np.ndarray) or (hasattr(queries, "shape") and not isinstance(queries, (list, tuple))):
                vectors = queries
            else:
                texts = list(queries)
                if any((not t or not t.strip()) for t in texts):
                    raise QueryError("Search query cannot be empty")
                vectors = np.asarray(await self.embedder.embed_batch(texts), dtype=np.float32)
            per_query = await self.qdrant.search_batch(collection=CollectionName.CODE_CHUNKS.value, query_vectors=vectors,
                                                       limit=limit, filters=filters or None)
        except EmbeddingError as e:
            raise QueryError("Failed to embed search query", cause=e)
        except VectorStoreError as e:
            raise QueryError("Failed to search code", cause=e)
        return [[_project(h, _CODE_KEYS) for h in hits] for hits in per_query]
