"""Embedding-provider plugin surface and the HIP UniXcoder provider.

* ``ProviderConfig`` / ``BaseEmbeddingProvider`` restate ``src/lattice/providers/base.py:21-64,138-225``:
  same constructor, ``embed`` (retried 5x, exponential back-off 1..60 s), ``embed_batch`` (sequential
  slices under a semaphore), ``set_concurrency`` and the overridable ``_embed_impl``.
* ``HipUniXcoderProvider`` replaces ``UniXcoderEmbeddingProvider``
  (``src/lattice/providers/unixcoder_provider.py:218-292``): same constructor and ``embedding_dim``, work
  offloaded to a 1-thread executor, but the encoder forward runs in hand-written HIP kernels
  (``encoder.HipUniXcoder``) and ragged batches are padded instead of failing (quirk Q1).
* ``get_embedding_provider`` restates the factory branch (``src/lattice/providers/factory.py:61-97,202-242``)
  for the provider names that are on this path.
"""

from __future__ import annotations

import asyncio
import logging
import os
from abc import ABC, abstractmethod
from collections import deque
from collections.abc import Sequence
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass, field

from .errors import ConfigurationError, EmbeddingError
from .settings import get_settings

logger = logging.getLogger(__name__)

RETRY_MAX_ATTEMPTS = 5
RETRY_MULTIPLIER = 1
RETRY_MIN_WAIT = 1
RETRY_MAX_WAIT = 60


class RetryError(Exception):
    """Raised by ``embed`` after the last attempt failed -- the reference's tenacity decorator (base.py:162-169,
    ``reraise`` left at its default) surfaces ``tenacity.RetryError`` there, not the original exception."""

    def __init__(self, last_exception: BaseException, attempts: int):
        super().__init__(f"RetryError after {attempts} attempts: {last_exception!r}")
        self.last_exception = last_exception
        self.attempts = attempts


def _backoff_seconds(attempt: int) -> float:
    """tenacity.wait_exponential(multiplier=1, min=1, max=60) after failed attempt number `attempt` (1-based)."""
    return float(max(RETRY_MIN_WAIT, min(RETRY_MULTIPLIER * 2 ** (attempt - 1), RETRY_MAX_WAIT)))


@dataclass
class ProviderConfig:
    provider: str
    model: str
    api_key: str | None = None
    base_url: str | None = None
    temperature: float = 0.7
    max_tokens: int = 1000
    extra: dict = field(default_factory=dict)

    @classmethod
    def from_env_prefix(cls, prefix: str) -> "ProviderConfig":
        return cls(
            provider=os.getenv(f"{prefix}_PROVIDER", "openai").lower(),
            model=os.getenv(f"{prefix}_MODEL", "gpt-4o"),
            api_key=os.getenv(f"{prefix}_API_KEY") or os.getenv("OPENAI_API_KEY"),
            base_url=os.getenv(f"{prefix}_BASE_URL"),
        )


class BaseEmbeddingProvider(ABC):
    # seconds -> awaitable; tests replace it so the back-off does not actually sleep
    _sleep = staticmethod(asyncio.sleep)

    def __init__(self, config: ProviderConfig):
        self.config = config
        self._semaphore = asyncio.Semaphore(5)

    @abstractmethod
    async def _embed_impl(self, texts: list[str]) -> list[list[float]]:
        """One vector per text, input order, Python floats."""

    async def embed(self, text: str) -> list[float]:
        """Every exception is retried, deterministic ones included (as the reference does)."""
        attempt = 0
        while True:
            attempt += 1
            try:
                return (await self._embed_batch_internal([text]))[0]
            except Exception as e:  # noqa: BLE001 - mirrors tenacity's catch-all
                if attempt >= RETRY_MAX_ATTEMPTS:
                    raise RetryError(e, attempt) from e
                await self._sleep(_backoff_seconds(attempt))

    async def _embed_batch_internal(self, texts: list[str]) -> list[list[float]]:
        async with self._semaphore:
            return await self._embed_impl(texts)

    async def embed_batch(self, texts: Sequence[str], batch_size: int = 100) -> list[list[float]]:
        items = list(texts)
        vectors: list[list[float]] = []
        for start in range(0, len(items), batch_size):
            vectors.extend(await self._embed_batch_internal(items[start:start + batch_size]))
        logger.debug(f"Generated {len(vectors)} embeddings")
        return vectors

    def set_concurrency(self, max_concurrent: int) -> None:
        self._semaphore = asyncio.Semaphore(max_concurrent)


class HipUniXcoderProvider(BaseEmbeddingProvider):
    """UniXcoder embeddings from the MI355X HIP encoder.  ``model`` may be a local checkpoint directory
    (``config.json`` + weights + ``vocab.json``/``merges.txt``); the hub name ``microsoft/unixcoder-base`` cannot be
    fetched offline, in which case ``CODERAG_HIP_WEIGHTS`` must point at a local copy, or
    ``extra={"synthetic_weights": seed}`` selects seeded random weights with the hashing tokenizer (benchmarks/tests)."""

    EMBEDDING_DIM = 768

    def __init__(self, config: ProviderConfig | None = None, max_length: int = 512):
        try:
            import torch  # noqa: F401
        except ImportError as e:  # unixcoder_provider.py:245-249
            raise RuntimeError("UniXcoder requires the 'torch' package (PyTorch-ROCm).") from e
        if config is None:
            config = ProviderConfig(provider="unixcoder-hip", model="microsoft/unixcoder-base")
        super().__init__(config)
        self.max_length = max_length
        self._executor = ThreadPoolExecutor(max_workers=1, thread_name_prefix="hip-unixcoder")
        self._model = None
        # cross-call dynamic batching: concurrent _embed_impl calls -- the orchestrator indexes 3
        # files at a time under a semaphore of 5 (pipeline/orchestrator.py:652-656, providers/base.py:148) -- are coalesced
        # into ONE length-bucketed GPU submission instead of queueing behind each other on the single worker thread
        self.dynamic_batching = bool(config.extra.get("dynamic_batching", True))
        # window 0 (default): no timer -- the calls that queue up while a submission is on the GPU travel in the next one, and an
        # idle provider embeds a lone call (a query) at once; > 0 additionally waits that long before each submission
        self.batch_window_s = float(config.extra.get("batch_window_ms", 0.0)) / 1e3
        self.max_batch_texts = int(config.extra.get("max_batch_texts", 4096))
        # "list" (default): list[list[float]] exactly as the reference's providers return; "numpy": list of float32 arrays --
        # what a store that converts to an array anyway (Qdrant's client does, HipVectorStore does) takes several times faster
        self.vector_rows = str(config.extra.get("vector_rows", "list"))
        self._pending: deque[tuple[list[str], asyncio.Future]] = deque()
        self._drainer: asyncio.Task | None = None
        self.submissions = 0                       # GPU submissions so far (observability / tests)
        logger.info("Initializing HIP UniXcoder embedding provider...")

    def _load(self):
        if self._model is None:
            from .encoder import load_unixcoder
            self._model = load_unixcoder(self.config.model, extra=self.config.extra)
        return self._model

    def _embed_sync(self, texts: list[str]) -> list[list[float]]:
        if not texts:
            return []
        try:
            model = self._load()
            kw = {} if self.vector_rows == "list" else {"rows": self.vector_rows}
            return model.embed_texts(texts, max_length=self.max_length, **kw)
        except Exception as e:
            raise EmbeddingError("HIP UniXcoder embedding failed", cause=e)

    def embed_texts_sync(self, texts: list[str]):
        """Blocking form for a caller that already sits on a worker thread -- ``HipVectorStore.upsert(vectors=None, texts=...,
        embed=provider.embed_texts_sync)``: the store routes the rows to their shards first and every process embeds only its
        own share.  Returns a float32 array [n, 768] (no Python float lists)."""
        import numpy as np
        if not texts:
            return np.zeros((0, self.EMBEDDING_DIM), np.float32)
        try:
            self.submissions += 1
            return self._load().embed_texts(list(texts), max_length=self.max_length, rows="array")
        except Exception as e:
            raise EmbeddingError("HIP UniXcoder embedding failed", cause=e)

    async def _embed_impl(self, texts: list[str]) -> list[list[float]]:
        loop = asyncio.get_running_loop()
        if not self.dynamic_batching:
            self.submissions += 1
            return await loop.run_in_executor(self._executor, self._embed_sync, list(texts))
        fut: asyncio.Future = loop.create_future()
        self._pending.append((list(texts), fut))
        if self._drainer is None or self._drainer.done():
            self._drainer = loop.create_task(self._drain())
        return await fut

    async def _drain(self) -> None:
        """Collect what arrives within the window (or until max_batch_texts), embed it in one submission, hand each caller
        its slice back in order.  A failure is delivered to every caller of that submission (each then retries on its own)."""
        loop = asyncio.get_running_loop()
        while self._pending:
            await asyncio.sleep(self.batch_window_s)
            batch, count = [], 0
            while self._pending and (not batch or count + len(self._pending[0][0]) <= self.max_batch_texts):
                item = self._pending.popleft()
                batch.append(item)
                count += len(item[0])
            flat = [t for texts, _ in batch for t in texts]
            try:
                self.submissions += 1
                vectors = await loop.run_in_executor(self._executor, self._embed_sync, flat)
                pos = 0
                for texts, fut in batch:
                    if not fut.done():
                        fut.set_result(vectors[pos:pos + len(texts)])
                    pos += len(texts)
            except Exception as e:  # noqa: BLE001
                for _, fut in batch:
                    if not fut.done():
                        fut.set_exception(e)

    @property
    def embedding_dim(self) -> int:
        return self.EMBEDDING_DIM

    def __del__(self):
        if hasattr(self, "_executor"):
            self._executor.shutdown(wait=False)


# the class name the reference's factory imports (factory.py:228-230)
UniXcoderEmbeddingProvider = HipUniXcoderProvider

_REMOTE = {"openai", "ollama", "google"}


def get_embedding_provider(provider: str | None = None, model: str | None = None, api_key: str | None = None,
                           base_url: str | None = None) -> BaseEmbeddingProvider:
    """Name -> provider.  ``unixcoder`` and ``unixcoder-hip`` both select the HIP encoder; the remote HTTP
    providers of the reference are outside this repo's scope and are reported as such."""
    settings = get_settings()
    name = (provider or settings.embedding_provider).lower()
    if name in ("unixcoder", "unixcoder-hip"):
        weights = model or settings.hip_weights or "microsoft/unixcoder-base"
        return HipUniXcoderProvider(ProviderConfig(provider=name, model=weights, api_key=api_key, base_url=base_url))
    if name in _REMOTE:
        raise ConfigurationError(f"Embedding provider '{name}' is a remote HTTP API and is not part of the MI355X hot path; "
                                 "use 'unixcoder' / 'unixcoder-hip' (or keep the reference's provider for it).")
    if name == "anthropic":
        raise ConfigurationError("Anthropic does not provide embedding models. "
                                 "Use 'openai', 'ollama', 'google', or 'unixcoder' for embeddings.")
    raise ConfigurationError(f"Unknown embedding provider: {name}. Supported providers: openai, ollama, google, unixcoder")
