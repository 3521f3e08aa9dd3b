"""Signal computation and the two score formulas of ``src/lattice/query/ranking/scorer.py:5-126``.

Every float expression keeps the reference's operand order (the products are summed left to right in
the order the reference writes them), so scores are identical to the last bit -- the golden file
``tests/golden/ranking_reference.json`` was produced by the reference's own code.
"""

from __future__ import annotations

from ..query_types import ResultSource
from .model import RankedResult, RankingConfig, RankingSignal

_S = RankingSignal


def _entity_match(name: str, query_entities: set[str]) -> float:
    """1.0 exact (case-folded) hit, 0.5 when any query entity is a substring, else 0 (scorer.py:31-35,92-96).
    An empty-string query entity is a substring of everything (quirk Q8) -- kept."""
    lowered = name.lower()
    if lowered in query_entities:
        return 1.0
    for qe in query_entities:
        if qe in lowered:
            return 0.5
    return 0.0


def _centrality(result: RankedResult, table: dict[str, dict[str, int]]) -> float:
    """min(1, total_degree / 50) looked up by qualified name, else entity name (scorer.py:48-54)."""
    degrees = table.get(result.qualified_name or result.entity_name)
    if degrees is None:
        return 0.0
    return min(1.0, degrees.get("total_degree", 0) / 50)


def _richness(result: RankedResult) -> float:
    score = 0.0
    for present, gain in ((result.summary, 0.3), (result.docstring, 0.2), (result.signature, 0.2), (result.content, 0.3)):
        if present:
            score += gain
    return score


def _quality(content: str | None) -> float:
    """Length heuristic of scorer.py:105-114."""
    if not content:
        return 0.0
    n = len(content)
    if 100 < n < 2000:
        return 0.8
    if 50 < n < 3000:
        return 0.5
    return 0.3


class ResultScorer:
    def __init__(self, config: RankingConfig):
        self.config = config

    def score_graph_result(self, result: RankedResult, weights: dict[str, float],
                           centrality_scores: dict[str, dict[str, int]], query_entities: set[str],
                           is_primary: bool = False, is_caller: bool = False, is_callee: bool = False) -> None:
        base = 1.0
        if not is_primary and (is_caller or is_callee):
            depth = result.depth_from_query or 1
            base = max(0.3, 1.0 - (depth - 1) * 0.2)
        relevance = 1.0 if is_primary else 0.8 if is_caller else 0.7 if is_callee else 0.5
        signals = {
            _S.GRAPH_MATCH.value: base,
            _S.QUERY_ENTITY_MATCH.value: _entity_match(result.entity_name, query_entities),
            _S.RELATIONSHIP_RELEVANCE.value: relevance,
            _S.CENTRALITY.value: _centrality(result, centrality_scores),
            _S.CONTEXT_RICHNESS.value: _richness(result),
        }
        result.final_score = (
            signals[_S.GRAPH_MATCH.value] * weights["graph_weight"]
            + signals[_S.QUERY_ENTITY_MATCH.value] * self.config.entity_match_bonus
            + signals[_S.RELATIONSHIP_RELEVANCE.value] * self.config.relationship_bonus
            + signals[_S.CENTRALITY.value] * weights["centrality_weight"]
            + signals[_S.CONTEXT_RICHNESS.value] * weights["context_weight"]
        )
        result.signal_scores = signals
        result.source = ResultSource.GRAPH.value

    def score_vector_result(self, result: RankedResult, vector_score: float, weights: dict[str, float],
                            centrality_scores: dict[str, dict[str, int]], query_entities: set[str]) -> None:
        signals = {
            _S.VECTOR_SIMILARITY.value: vector_score,
            _S.QUERY_ENTITY_MATCH.value: _entity_match(result.entity_name, query_entities),
            _S.CENTRALITY.value: _centrality(result, centrality_scores),
            _S.CODE_QUALITY.value: _quality(result.content),
        }
        result.final_score = (
            signals[_S.VECTOR_SIMILARITY.value] * weights["vector_weight"]
            + signals[_S.QUERY_ENTITY_MATCH.value] * self.config.entity_match_bonus
            + signals[_S.CENTRALITY.value] * weights["centrality_weight"]
            + signals[_S.CODE_QUALITY.value] * 0.1
        )
        result.signal_scores = signals
        result.source = ResultSource.VECTOR.value
