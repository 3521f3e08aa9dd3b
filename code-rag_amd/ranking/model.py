"""Ranking data types: same names, fields and defaults as ``src/lattice/query/ranking/models.py:8-91``."""

from __future__ import annotations

from dataclasses import dataclass, field
from enum import Enum
from typing import Any

from ..query_types import QueryIntent, ResultSource, intent_key

DEFAULT_GRAPH_WEIGHT = 0.5
DEFAULT_VECTOR_WEIGHT = 0.5
DEFAULT_CENTRALITY_WEIGHT = 0.2
DEFAULT_CONTEXT_WEIGHT = 0.1
MAX_RESULTS_PER_FILE = 5
MAX_TOTAL_RESULTS = 50


class RankingSignal(Enum):
    GRAPH_MATCH = "graph_match"
    VECTOR_SIMILARITY = "vector_similarity"
    CENTRALITY = "centrality"
    QUERY_ENTITY_MATCH = "query_entity_match"
    RELATIONSHIP_RELEVANCE = "relationship_relevance"
    CODE_QUALITY = "code_quality"
    CONTEXT_RICHNESS = "context_richness"


@dataclass
class RankedResult:
    file_path: str
    entity_name: str
    entity_type: str
    qualified_name: str | None = None
    content: str | None = None
    summary: str | None = None
    signature: str | None = None
    docstring: str | None = None
    start_line: int | None = None
    end_line: int | None = None
    source: str = ResultSource.HYBRID.value
    graph_node_id: str | None = None
    final_score: float = 0.0
    signal_scores: dict[str, float] = field(default_factory=dict)
    callers: list[str] = field(default_factory=list)
    callees: list[str] = field(default_factory=list)
    depth_from_query: int | None = None
    relationship_path: str | None = None
    metadata: dict[str, Any] = field(default_factory=dict)

    def get_key(self) -> str:
        """Merge / dedup key (models.py:55-56): file, entity name, start line."""
        return f"{self.file_path}:{self.entity_name}:{self.start_line}"


# (graph_weight, vector_weight) per intent -- the 14 rows of models.py:77-90; intents not listed keep the defaults
_INTENT_WEIGHTS: dict[QueryIntent, tuple[float, float]] = {
    QueryIntent.FIND_CALLERS: (0.8, 0.2),
    QueryIntent.FIND_CALLEES: (0.8, 0.2),
    QueryIntent.FIND_CALL_CHAIN: (0.9, 0.1),
    QueryIntent.FIND_HIERARCHY: (0.85, 0.15),
    QueryIntent.FIND_USAGES: (0.7, 0.3),
    QueryIntent.FIND_DEPENDENCIES: (0.75, 0.25),
    QueryIntent.LOCATE_ENTITY: (0.6, 0.4),
    QueryIntent.LOCATE_FILE: (0.5, 0.5),
    QueryIntent.EXPLAIN_IMPLEMENTATION: (0.5, 0.5),
    QueryIntent.EXPLAIN_RELATIONSHIP: (0.6, 0.4),
    QueryIntent.EXPLAIN_DATA_FLOW: (0.65, 0.35),
    QueryIntent.FIND_SIMILAR: (0.2, 0.8),
    QueryIntent.SEARCH_FUNCTIONALITY: (0.3, 0.7),
    QueryIntent.SEARCH_PATTERN: (0.25, 0.75),
}


@dataclass
class RankingConfig:
    graph_weight: float = DEFAULT_GRAPH_WEIGHT
    vector_weight: float = DEFAULT_VECTOR_WEIGHT
    centrality_weight: float = DEFAULT_CENTRALITY_WEIGHT
    context_weight: float = DEFAULT_CONTEXT_WEIGHT
    entity_match_bonus: float = 0.3
    relationship_bonus: float = 0.15
    query_type_adjustments: dict[Any, dict[str, float]] = field(default_factory=dict)
    max_per_file: int = MAX_RESULTS_PER_FILE
    max_total: int = MAX_TOTAL_RESULTS

    def __post_init__(self) -> None:
        if not self.query_type_adjustments:
            self.query_type_adjustments = {
                intent: {"graph_weight": g, "vector_weight": v} for intent, (g, v) in _INTENT_WEIGHTS.items()
            }

    def weights_for(self, intent: Any) -> dict[str, float]:
        """Base weights overridden by the intent's row (ranker.py:56-68).  The lookup goes through the intent's
        string value so a plan built with the reference's own ``QueryIntent`` enum selects the same row."""
        weights = {"graph_weight": self.graph_weight, "vector_weight": self.vector_weight,
                   "centrality_weight": self.centrality_weight, "context_weight": self.context_weight}
        wanted = intent_key(intent)
        for key, override in self.query_type_adjustments.items():
            if intent_key(key) == wanted:
                weights.update(override)
                break
        return weights
