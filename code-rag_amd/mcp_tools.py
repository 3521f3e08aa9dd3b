"""MCP ``semantic_search`` tool -- ``src/lattice/mcp/tools.py:368-462`` with its two wiring faults (quirk Q4) repaired:
the searcher comes from a factory that really constructs it, and code hits carry ``summary=None`` instead of raising."""

from __future__ import annotations

import logging
from dataclasses import dataclass
from typing import Any, Callable

logger = logging.getLogger(__name__)


@dataclass
class ToolResult:
    success: bool
    data: Any = None
    message: str | None = None
    error: str | None = None


@dataclass
class SearchResult:
    qualified_name: str
    entity_type: str
    file_path: str
    score: float
    summary: str | None = None


def create_semantic_search_tool(vector_searcher_factory: Callable[[], Any]) -> dict[str, Any]:
    async def semantic_search(query: str, limit: int = 5, entity_type: str | None = None) -> ToolResult:
        logger.info(f"[Tool:SemanticSearch] Query: '{query}'")
        try:
            searcher = vector_searcher_factory()
            hits = await searcher.search_code(query=query, limit=limit, entity_type=entity_type)
            rows = []
            for h in hits:
                get = h.get if isinstance(h, dict) else (lambda k, _h=h: getattr(_h, k, None))
                rows.append(SearchResult(qualified_name=get("entity_name"), entity_type=get("entity_type"),
                                         file_path=get("file_path"), score=get("score"), summary=get("summary")))
            return ToolResult(success=True, data=[vars(r) for r in rows], message=f"Found {len(rows)} matches for '{query}'.")
        except Exception as e:  # the reference's catch-all (tools.py:431-436)
            logger.error(f"[Tool:SemanticSearch] Error: {e}", exc_info=True)
            return ToolResult(success=False, error=str(e))

    return {
        "name": "semantic_search",
        "description": ("Search for code by functionality or intent using natural language. "
                        "Find code based on what it does, not its name."),
        "function": semantic_search,
        "parameters": {
            "query": {"type": "string", "description": "Natural language description of functionality", "required": True},
            "limit": {"type": "integer", "description": "Maximum number of results (default: 5)", "required": False},
            "entity_type": {"type": "string", "description": "Filter by type: function, class, method", "required": False},
        },
    }
