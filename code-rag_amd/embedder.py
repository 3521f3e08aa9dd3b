"""``Embedder`` facade -- restates ``src/lattice/embeddings/embedder.py:11-73`` over the HIP provider."""

from __future__ import annotations

import logging
from collections.abc import Callable, Sequence

from .providers import BaseEmbeddingProvider, get_embedding_provider
from .settings import get_settings

logger = logging.getLogger(__name__)


class Embedder:
    def __init__(self, provider: str | None = None, model: str | None = None, api_key: str | None = None,
                 base_url: str | None = None, max_concurrent: int | None = None, *,
                 provider_instance: BaseEmbeddingProvider | None = None):
        self.max_concurrent = max_concurrent or get_settings().max_concurrent_requests
        self._provider = provider_instance or get_embedding_provider(provider=provider, model=model, api_key=api_key,
                                                                     base_url=base_url)
        self._provider.set_concurrency(self.max_concurrent)
        logger.info(f"Initialized Embedder with {self._provider.config.provider}/{self._provider.config.model}")

    @property
    def embedding_dim(self) -> int | None:
        return getattr(self._provider, "embedding_dim", None)

    async def embed(self, text: str) -> list[float]:
        return await self._provider.embed(text)

    async def embed_batch(self, texts: Sequence[str], batch_size: int = 100) -> list[list[float]]:
        return await self._provider.embed_batch(texts, batch_size)

    async def embed_with_progress(self, texts: Sequence[str], batch_size: int = 100,
                                  progress_callback: Callable[[int, int], None] | None = None) -> list[list[float]]:
        """Slices of ``batch_size``; after each one ``progress_callback(min(done, total), total)`` (embedder.py:48-70)."""
        items = list(texts)
        total = len(items)
        vectors: list[list[float]] = []
        for start in range(0, total, batch_size):
            part = items[start:start + batch_size]
            vectors.extend(await self._provider.embed_batch(part, batch_size=len(part)))
            if progress_callback:
                progress_callback(min(start + batch_size, total), total)
        logger.info(f"Generated {len(vectors)} embeddings across {(total + batch_size - 1) // batch_size} batches")
        return vectors

    async def embed_array(self, texts: Sequence[str]):
        """All texts in ONE coalesced submission, returned as a float32 array [n, dim] -- no Python float lists between the
        encoder and the store (the reference's ``list[list[float]]`` costs as much as the GPU forward at this size).  Not in the
        reference; ``VectorIndexer.index_files_batched`` uses it.  A provider without an array path goes through
        :meth:`embed_batch`."""
        import asyncio
        import numpy as np
        items = list(texts)
        fn = getattr(self._provider, "embed_texts_sync", None)
        if fn is None:
            return np.asarray(await self._provider.embed_batch(items, batch_size=max(1, len(items))), dtype=np.float32)
        loop = asyncio.get_running_loop()
        executor = getattr(self._provider, "_executor", None)          # the provider's own worker thread: one thread drives the encoder
        return await loop.run_in_executor(executor, fn, items)

    @property
    def provider(self) -> BaseEmbeddingProvider:
        return self._provider


OpenAIEmbedder = Embedder  # alias kept by the reference (embedder.py:73)
