"""HipVectorStore's QdrantManager surface (src/lattice/embeddings/client.py:18-228) -- host-side behaviour, with the
device index replaced by the oracle-backed FakeIndex.  The same scenarios run against the real index in
tests/test_store_gpu.py."""
import asyncio

import numpy as np
import pytest

import coderag_amd  # noqa: F401
from coderag_amd import ffi, store as store_mod
from coderag_amd.errors import VectorStoreError
from tests.fake_index import FakeIndex
from tests.store_scenarios import run_store_scenarios


@pytest.fixture
def fake(monkeypatch):
    monkeypatch.setattr(ffi, "Index", FakeIndex)
    monkeypatch.setattr(ffi, "lib", lambda: object())
    monkeypatch.setattr(ffi, "device_count", lambda: 1)
    monkeypatch.setattr(ffi, "device_info", lambda d=0: {"name": "fake", "arch": "gfx950", "hbm_bytes": 0, "cu_count": 256})
    return FakeIndex


def test_store_scenarios_on_fake_index(fake):
    asyncio.run(run_store_scenarios(store_mod.HipVectorStore(dim=768, initial_capacity=64)))


def test_client_before_connect_raises(fake):
    s = store_mod.HipVectorStore(dim=768)
    with pytest.raises(VectorStoreError, match="Client not connected"):
        _ = s.client

    async def go():
        with pytest.raises(VectorStoreError):
            await s.create_collections()
        assert await s.health_check() is False
        assert await s.file_needs_update("code_chunks", "a.py", "h") is True      # True on ANY error
    asyncio.run(go())


def test_connect_failure_is_wrapped(monkeypatch):
    monkeypatch.setattr(ffi, "device_count", lambda: 0)
    s = store_mod.HipVectorStore(dim=768)

    async def go():
        with pytest.raises(VectorStoreError, match="Failed to connect"):
            await s.connect()
    asyncio.run(go())


def test_dimension_defaults_follow_provider(monkeypatch):
    monkeypatch.setenv("EMBEDDING_PROVIDER", "unixcoder")
    assert store_mod.HipVectorStore()._dimensions == 768
    monkeypatch.setenv("EMBEDDING_PROVIDER", "openai")
    monkeypatch.setenv("EMBEDDING_DIMENSIONS", "1536")
    assert store_mod.HipVectorStore()._dimensions == 1536         # quirk Q3 reproduced unless dim= is passed
    assert store_mod.QdrantManager is store_mod.HipVectorStore
    assert [c.value for c in store_mod.CollectionName] == ["code_chunks", "summaries"]
