"""Input types of the ranking stage.

The ranker consumes a query plan and a graph context that the reference builds elsewhere (LLM planner,
Memgraph traversals -- both out of scope).  Only their data shapes are on the hot path; they are
re-declared here with the reference's field names so plans/contexts built by the reference can be passed
in unchanged (duck typing) and so tests can construct them:

* ``QueryIntent``, ``ExtractedEntity``, ``QueryPlan``  <- src/lattice/query/query_planner.py:24-91
* ``GraphNode``, ``GraphContext``                      <- src/lattice/query/graph_reasoning/models.py:17-54
* ``ResultSource``                                     <- src/lattice/core/types.py:52-55
"""

from __future__ import annotations

from dataclasses import dataclass, field
from enum import Enum
from typing import Any


class ResultSource(str, Enum):
    GRAPH = "graph"
    VECTOR = "vector"
    HYBRID = "hybrid"


class QueryIntent(Enum):
    # structural
    FIND_CALLERS = "find_callers"
    FIND_CALLEES = "find_callees"
    FIND_CALL_CHAIN = "find_call_chain"
    FIND_HIERARCHY = "find_hierarchy"
    FIND_IMPLEMENTATIONS = "find_implementations"
    FIND_USAGES = "find_usages"
    FIND_DEPENDENCIES = "find_dependencies"
    FIND_DEPENDENTS = "find_dependents"
    # navigational
    LOCATE_ENTITY = "locate_entity"
    LOCATE_FILE = "locate_file"
    # explanatory
    EXPLAIN_IMPLEMENTATION = "explain_implementation"
    EXPLAIN_RELATIONSHIP = "explain_relationship"
    EXPLAIN_DATA_FLOW = "explain_data_flow"
    EXPLAIN_ARCHITECTURE = "explain_architecture"
    # semantic
    FIND_SIMILAR = "find_similar"
    SEARCH_FUNCTIONALITY = "search_functionality"
    SEARCH_PATTERN = "search_pattern"


@dataclass
class ExtractedEntity:
    name: str
    entity_type: str | None = None
    is_primary: bool = False
    context: str | None = None


@dataclass
class QueryPlan:
    """Only ``primary_intent`` and ``entities`` are read by the ranker (ranking/ranker.py:28,75)."""

    original_query: str
    primary_intent: QueryIntent
    sub_queries: list[Any] = field(default_factory=list)
    entities: list[ExtractedEntity] = field(default_factory=list)
    relationships: list[Any] = field(default_factory=list)
    requires_multi_hop: bool = False
    max_hops: int = 1
    context_requirements: list[str] = field(default_factory=list)
    reasoning: str = ""


@dataclass
class GraphNode:
    node_type: str
    name: str
    qualified_name: str
    file_path: str
    signature: str | None = None
    docstring: str | None = None
    summary: str | None = None
    start_line: int | None = None
    end_line: int | None = None
    is_async: bool = False
    parent_class: str | None = None
    metadata: dict[str, Any] = field(default_factory=dict)


@dataclass
class GraphContext:
    primary_entities: list[GraphNode] = field(default_factory=list)
    callers: list[GraphNode] = field(default_factory=list)
    callees: list[GraphNode] = field(default_factory=list)
    parent_classes: list[GraphNode] = field(default_factory=list)
    child_classes: list[GraphNode] = field(default_factory=list)
    methods: list[GraphNode] = field(default_factory=list)
    containing_class: GraphNode | None = None
    file_context: list[GraphNode] = field(default_factory=list)
    dependencies: list[GraphNode] = field(default_factory=list)
    dependents: list[GraphNode] = field(default_factory=list)
    call_chains: list[Any] = field(default_factory=list)
    inheritance_chains: list[Any] = field(default_factory=list)


def intent_key(intent: Any) -> str:
    """Intent -> its string value, for enums from this module, from the reference, or plain strings."""
    return getattr(intent, "value", intent)
