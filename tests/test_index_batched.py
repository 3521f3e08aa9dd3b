"""``VectorIndexer.index_files_batched``: the reference's indexing flow (src/lattice/embeddings/indexer.py:46-119 --
per file: hash check, delete by path, chunk, embed, upsert) as ONE pass over all files.  It must leave the store exactly as
the sequential flow does (same chunks, payloads, vectors; ids are fresh uuid4s either way) while asking the store and the
embedder once instead of once per file.  CPU tier: the oracle-backed FakeIndex under HipVectorStore, a deterministic provider."""
import asyncio
import types
import zlib
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
import pytest

import coderag_amd  # noqa: F401
from coderag_amd import ffi, store as store_mod
from coderag_amd.embedder import Embedder
from coderag_amd.indexer import CodeChunker, VectorIndexer
from coderag_amd.providers import BaseEmbeddingProvider, ProviderConfig
from tests.fake_index import FakeIndex


@pytest.fixture
def fake(monkeypatch):
    monkeypatch.setattr(ffi, "Index", FakeIndex)
    monkeypatch.setattr(ffi, "lib", lambda: object())
    monkeypatch.setattr(ffi, "device_count", lambda: 1)
    monkeypatch.setattr(ffi, "device_info", lambda d=0: {"name": "fake", "arch": "gfx950", "hbm_bytes": 0, "cu_count": 256})


def _vec(text):
    return np.random.default_rng(zlib.crc32(text.encode())).standard_normal(768).astype(np.float32)


class CountingProvider(BaseEmbeddingProvider):
    """Text -> a vector that depends on the text alone; counts how often the 'encoder' was driven."""
    embedding_dim = 768

    def __init__(self, with_array_path=True):
        super().__init__(ProviderConfig(provider="fake", model="fake"))
        self.impl_calls, self.sync_calls = [], []
        self._executor = ThreadPoolExecutor(max_workers=1)
        if not with_array_path:
            self.embed_texts_sync = None

    async def _embed_impl(self, texts):
        self.impl_calls.append(len(texts))
        return [_vec(t).tolist() for t in texts]

    def embed_texts_sync(self, texts):
        self.sync_calls.append(len(texts))
        return np.stack([_vec(t) for t in texts]) if texts else np.zeros((0, 768), np.float32)


def _parsed_file(i, n_funcs, content_hash="h1", salt=""):
    ents = [types.SimpleNamespace(type=types.SimpleNamespace(value="function"), name=f"fn_{i}_{j}", qualified_name=f"mod{i}.fn_{i}_{j}",
                                  signature=f"def fn_{i}_{j}(x)", docstring="Does things.", code=f"    return x * {j} + {i}{salt}\n" * (1 + j % 3),
                                  start_line=10 * j + 1, end_line=10 * j + 8) for j in range(n_funcs)]
    info = types.SimpleNamespace(path=Path(f"/proj/mod{i}.py"), content_hash=content_hash, language=types.SimpleNamespace(value="python"))
    return types.SimpleNamespace(file_info=info, content="", all_entities=ents)


async def _contents(store):
    """Everything the store holds, by (file, entity): payload + the vector's effect (its best hit is itself)."""
    out = {}
    for f in range(12):
        for h in await store.search("code_chunks", None, limit=100, filters={"file_path": f"/proj/mod{f}.py"}):
            out[(h["payload"]["file_path"], h["payload"]["entity_name"])] = h["payload"]
    return out


def test_batched_indexing_leaves_the_store_as_the_sequential_flow_does(fake):
    files = [_parsed_file(i, 3 + i % 4) for i in range(12)]

    async def run(batched, provider):
        async with store_mod.HipVectorStore(dim=768, initial_capacity=64) as store:
            await store.create_collections()
            indexer = VectorIndexer(store, Embedder(provider_instance=provider), CodeChunker(max_tokens=1000, overlap_tokens=200))
            prog = []
            fn = indexer.index_files_batched if batched else indexer.index_files
            n = await fn(files, progress_callback=lambda d, t: prog.append((d, t)), project_name="proj")
            first = await _contents(store)
            # second run, nothing changed: every file is skipped
            assert await fn(files, project_name="proj") == 0
            # third run: two files changed (new hash, new code), one file new
            changed = [_parsed_file(2, 5, "h2", salt=" + 1"), _parsed_file(7, 2, "h2", salt=" - 1"), _parsed_file(40, 3)]
            m = await fn(files[:2] + changed, project_name="proj")
            third = await _contents(store)
            third.update({k: v for f in (40,) for k, v in [((h["payload"]["file_path"], h["payload"]["entity_name"]), h["payload"])
                                                             for h in await store.search("code_chunks", None, limit=100, filters={"file_path": f"/proj/mod{f}.py"})]})
            # a stored chunk is found by its own text's vector, under the payload it was stored with
            probe = files[5].all_entities[1]
            text = "\n".join([probe.signature, f'"""{probe.docstring}"""', probe.code])
            top = (await store.search("code_chunks", _vec(text).tolist(), limit=1))[0]
            count = (await store.get_collection_info("code_chunks")).points_count
            return n, prog, first, m, third, (top["payload"]["entity_name"], round(top["score"], 5)), count
    seq_p, bat_p = CountingProvider(), CountingProvider()
    seq = asyncio.run(run(False, seq_p))
    bat = asyncio.run(run(True, bat_p))
    assert seq[0] == bat[0] == sum(3 + i % 4 for i in range(12)) and seq[1] == bat[1] == [(d, 12) for d in range(1, 13)]
    assert seq[2] == bat[2] and seq[3] == bat[3] == 5 + 2 + 3 and seq[4] == bat[4] and seq[5] == bat[5] and seq[6] == bat[6]
    assert seq[5][0] == "mod5.fn_5_1" and seq[5][1] == 1.0
    # the sequential flow drives the encoder once per file (12 + 3 submissions); the batched one once per CALL that has work
    assert len(seq_p.impl_calls) == 15 and sum(seq_p.impl_calls) == seq[0] + 10
    assert bat_p.sync_calls == [bat[0], 10] and bat_p.impl_calls == []


def test_batched_indexing_without_an_array_path_and_with_a_failing_batch(fake):
    """A provider that only has the reference's list surface still works (one embed_batch call); when the batch as a whole
    fails the files go through the sequential flow, where a bad file loses only itself (indexer.py:110-113)."""
    files = [_parsed_file(i, 2) for i in range(4)]

    async def go():
        async with store_mod.HipVectorStore(dim=768, initial_capacity=64) as store:
            await store.create_collections()
            p = CountingProvider(with_array_path=False)
            indexer = VectorIndexer(store, Embedder(provider_instance=p), CodeChunker())
            assert await indexer.index_files_batched(files, project_name="proj") == 8 and p.impl_calls == [8]

            class Boom(CountingProvider):
                def embed_texts_sync(self, texts):
                    if any("fn_2_" in t for t in texts):
                        raise RuntimeError("bad text")
                    return super().embed_texts_sync(texts)

                async def _embed_impl(self, texts):
                    if any("fn_2_" in t for t in texts):
                        raise RuntimeError("bad text")
                    return await super()._embed_impl(texts)
            b = Boom()
            b._sleep = staticmethod(lambda s: asyncio.sleep(0))
            indexer = VectorIndexer(store, Embedder(provider_instance=b), CodeChunker())
            n = await indexer.index_files_batched(files, project_name="proj", force=True)
            assert n == 6                                                    # files 0, 1, 3 went through; file 2 was logged and skipped
            assert (await store.get_collection_info("code_chunks")).points_count == 6 + 0
    asyncio.run(go())
