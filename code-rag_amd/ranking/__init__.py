"""Hybrid graph/vector re-rank (reference: ``src/lattice/query/ranking/``)."""
from .hybrid import HybridRanker, ranked_results_to_search_results
from .model import (DEFAULT_CENTRALITY_WEIGHT, DEFAULT_CONTEXT_WEIGHT, DEFAULT_GRAPH_WEIGHT, DEFAULT_VECTOR_WEIGHT,
                    MAX_RESULTS_PER_FILE, MAX_TOTAL_RESULTS, RankedResult, RankingConfig, RankingSignal)
from .scoring import ResultScorer

__all__ = ["DEFAULT_CENTRALITY_WEIGHT", "DEFAULT_CONTEXT_WEIGHT", "DEFAULT_GRAPH_WEIGHT", "DEFAULT_VECTOR_WEIGHT",
           "HybridRanker", "MAX_RESULTS_PER_FILE", "MAX_TOTAL_RESULTS", "RankedResult", "RankingConfig", "RankingSignal",
           "ResultScorer", "ranked_results_to_search_results"]
