"""``HybridRanker``: fuse graph-traversal results and vector hits into one ranked list.

Behavioural restatement of ``src/lattice/query/ranking/ranker.py:13-264`` and ``utils.py:6-30``:
graph nodes are scored role by role (primary, caller, callee, method, parent, child -- in that order),
vector hits after them; entries sharing ``file:entity:start_line`` are merged ((a+b)/2 * 1.1, per-signal
max, source "hybrid"); the list is sorted by score (stable: graph before vector, then store order),
capped at 5 per file and 50 in total.
"""

from __future__ import annotations

import logging
from typing import Any

from ..query_types import ResultSource
from .model import RankedResult, RankingConfig
from .scoring import ResultScorer

logger = logging.getLogger(__name__)

# graph role -> (GraphContext attribute, relationship_path, scorer flags, carries a traversal depth)
_GRAPH_ROLES = (
    ("primary_entities", None, {"is_primary": True}, False),
    ("callers", "caller", {"is_caller": True}, True),
    ("callees", "callee", {"is_callee": True}, True),
    ("methods", "method", {}, False),
    ("parent_classes", "parent_class", {}, False),
    ("child_classes", "child_class", {}, False),
)
_FILL_IF_MISSING = ("content", "summary", "signature", "docstring")


class HybridRanker:
    def __init__(self, config: RankingConfig | None = None):
        self.config = config or RankingConfig()
        self.scorer = ResultScorer(self.config)

    def rank_results(self, plan, graph_context, vector_results: list[dict[str, Any]],
                     centrality_scores: dict[str, dict[str, int]] | None = None) -> list[RankedResult]:
        weights = self.config.weights_for(plan.primary_intent)
        centrality = centrality_scores or {}
        wanted = {e.name.lower() for e in plan.entities}
        merged: dict[str, RankedResult] = {}

        for attr, path, flags, has_depth in _GRAPH_ROLES:
            for node in getattr(graph_context, attr):
                cand = self._from_graph_node(node)
                if path is not None:
                    cand.relationship_path = path
                if has_depth:
                    cand.depth_from_query = node.metadata.get("depth", 1) if node.metadata else 1
                self.scorer.score_graph_result(cand, weights, centrality, wanted, **flags)
                self._absorb(merged, cand)

        for hit in vector_results:
            cand = self._from_vector_hit(hit)
            self.scorer.score_vector_result(cand, hit.get("score", 0.0), weights, centrality, wanted)
            self._absorb(merged, cand)

        ordered = list(merged.values())
        ordered.sort(key=lambda r: r.final_score, reverse=True)
        final = self._cap(ordered)
        logger.debug("Ranking complete: %d results after deduplication", len(final))
        return final

    # ranker.py:171-202
    @staticmethod
    def _absorb(merged: dict[str, RankedResult], cand: RankedResult) -> None:
        key = cand.get_key()
        held = merged.get(key)
        if held is None:
            merged[key] = cand
            return
        combined = (held.final_score + cand.final_score) / 2
        combined *= 1.1
        for name in _FILL_IF_MISSING:
            if not getattr(held, name) and getattr(cand, name):
                setattr(held, name, getattr(cand, name))
        for signal, value in cand.signal_scores.items():
            held.signal_scores[signal] = max(held.signal_scores[signal], value) if signal in held.signal_scores else value
        held.final_score = combined
        held.source = ResultSource.HYBRID.value

    # ranker.py:204-229
    def _cap(self, ordered: list[RankedResult]) -> list[RankedResult]:
        kept: list[RankedResult] = []
        seen: set[str] = set()
        per_file: dict[str, int] = {}
        for r in ordered:
            key = r.get_key()
            if key in seen or per_file.get(r.file_path, 0) >= self.config.max_per_file:
                continue
            seen.add(key)
            per_file[r.file_path] = per_file.get(r.file_path, 0) + 1
            kept.append(r)
            if len(kept) >= self.config.max_total:
                break
        return kept

    @staticmethod
    def _from_graph_node(node) -> RankedResult:
        return RankedResult(file_path=node.file_path, entity_name=node.name, entity_type=node.node_type,
                            qualified_name=node.qualified_name, summary=node.summary, signature=node.signature,
                            docstring=node.docstring, start_line=node.start_line, end_line=node.end_line,
                            graph_node_id=node.qualified_name, metadata=node.metadata)

    @staticmethod
    def _from_vector_hit(hit: dict[str, Any]) -> RankedResult:
        return RankedResult(file_path=hit.get("file_path", ""), entity_name=hit.get("entity_name", ""),
                            entity_type=hit.get("entity_type", ""), qualified_name=hit.get("graph_node_id"),
                            content=hit.get("content"), summary=hit.get("summary"), start_line=hit.get("start_line"),
                            end_line=hit.get("end_line"), graph_node_id=hit.get("graph_node_id"))


def ranked_results_to_search_results(results: list[RankedResult]) -> list[dict[str, Any]]:
    """Flatten to the dict shape ``QueryEngine.search`` returns (ranking/utils.py:6-30)."""
    flat = []
    for r in results:
        flat.append({
            "source": r.source, "score": r.final_score, "file_path": r.file_path, "entity_type": r.entity_type,
            "entity_name": r.entity_name, "content": r.content, "summary": r.summary, "start_line": r.start_line,
            "end_line": r.end_line, "graph_node_id": r.graph_node_id,
            "metadata": {"signal_scores": r.signal_scores, "relationship_path": r.relationship_path,
                         "depth_from_query": r.depth_from_query, "signature": r.signature, "docstring": r.docstring,
                         "callers": r.callers, "callees": r.callees},
        })
    return flat
