"""Environment-driven settings of the hot path.

Reads the same flat env names as the reference's pydantic-settings model
(``src/lattice/config/settings.py:44-55,73-76,132-134``) without importing it, plus this repo's
own ``CODERAG_HIP_*`` knobs.  Values are read at call time (no caching) so tests can monkeypatch.
"""

from __future__ import annotations

import os
from dataclasses import dataclass


def _int(name: str, default: int) -> int:
    raw = os.environ.get(name)
    return default if raw in (None, "") else int(raw)


@dataclass(frozen=True)
class HotPathSettings:
    embedding_provider: str
    embedding_model: str
    embedding_dimensions: int
    max_concurrent_requests: int
    batch_size: int
    search_limit: int
    max_vector_results: int
    max_centrality_lookups: int
    hip_device: int
    hip_store_dtype: str
    hip_initial_capacity: int
    hip_weights: str | None
    hip_shards: int
    hip_shard_backend: str
    hip_compact_dead_fraction: float
    hip_compact_min_dead: int
    hip_store_stream: str


def get_settings() -> HotPathSettings:
    return HotPathSettings(
        embedding_provider=os.environ.get("EMBEDDING_PROVIDER", "openai").lower(),
        embedding_model=os.environ.get("EMBEDDING_MODEL", "text-embedding-3-small"),
        embedding_dimensions=_int("EMBEDDING_DIMENSIONS", 1536),       # settings.py:53 (UniXcoder needs 768: quirk Q3)
        max_concurrent_requests=_int("MAX_CONCURRENT_REQUESTS", 5),    # settings.py:74
        batch_size=_int("BATCH_SIZE", 100),
        search_limit=_int("SEARCH_LIMIT", 15),                         # settings.py:132
        max_vector_results=_int("MAX_VECTOR_RESULTS", 20),             # settings.py:133
        max_centrality_lookups=_int("MAX_CENTRALITY_LOOKUPS", 10),     # settings.py:134
        hip_device=_int("CODERAG_HIP_DEVICE", 0),
        hip_store_dtype=os.environ.get("CODERAG_HIP_STORE_DTYPE", "f32").lower(),
        hip_initial_capacity=_int("CODERAG_HIP_INITIAL_CAPACITY", 65536),
        hip_weights=os.environ.get("CODERAG_HIP_WEIGHTS") or None,
        hip_shards=_int("CODERAG_HIP_SHARDS", 1),                      # row shards per collection (store.py)
        hip_shard_backend=os.environ.get("CODERAG_HIP_SHARD_BACKEND", "auto").lower(),     # local | dist | auto
        hip_compact_dead_fraction=float(os.environ.get("CODERAG_HIP_COMPACT_DEAD_FRACTION", "0.25") or 0.25),
        hip_compact_min_dead=_int("CODERAG_HIP_COMPACT_MIN_DEAD", 1024),
        hip_store_stream=os.environ.get("CODERAG_HIP_STORE_STREAM", "priority").lower(),    # priority | default (store.py: the stream its device work runs on)
    )
