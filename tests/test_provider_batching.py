"""HipUniXcoderProvider host logic on CPU (the encoder is replaced by a recording stub): cross-call dynamic batching,
order preservation, error fan-out, the reference's constructor/factory surface."""
import asyncio

import pytest

import coderag_amd  # noqa: F401
from coderag_amd import providers as P
from coderag_amd.errors import ConfigurationError, EmbeddingError


class StubModel:
    def __init__(self):
        self.calls = []

    def embed_texts(self, texts, max_length=512):
        self.calls.append(list(texts))
        if any(t == "poison" for t in texts):
            raise RuntimeError("kernel fault")
        return [[float(len(t)), float(sum(map(ord, t)) % 997)] for t in texts]


def make(extra=None):
    p = P.HipUniXcoderProvider(P.ProviderConfig(provider="unixcoder-hip", model="stub", extra=extra or {}))
    p._model = StubModel()

    async def no_sleep(sec):
        return None
    p._sleep = no_sleep
    return p


def test_concurrent_calls_share_one_submission():
    p = make({"batch_window_ms": 5})

    async def go():
        groups = [[f"file{f}_chunk{c}" for c in range(3 + f)] for f in range(4)]       # 4 "files" embedded concurrently
        outs = await asyncio.gather(*(p.embed_batch(g, batch_size=100) for g in groups), p.embed("single query"))
        return groups, outs
    groups, outs = asyncio.run(go())
    assert p.submissions == 1 and len(p._model.calls) == 1 and len(p._model.calls[0]) == sum(map(len, groups)) + 1
    for g, out in zip(groups, outs[:-1]):                                             # each caller gets ITS vectors, in order
        assert out == [[float(len(t)), float(sum(map(ord, t)) % 997)] for t in g]
    assert outs[-1] == [float(len("single query")), float(sum(map(ord, "single query")) % 997)]


def test_batching_can_be_disabled_and_respects_max_batch():
    p = make({"dynamic_batching": False})

    async def go():
        await asyncio.gather(p.embed_batch(["a", "b"]), p.embed_batch(["c"]))
    asyncio.run(go())
    assert p.submissions == 2 and sorted(map(len, p._model.calls)) == [1, 2]
    p2 = make({"batch_window_ms": 1, "max_batch_texts": 3})

    async def go2():
        return await asyncio.gather(*(p2.embed_batch([f"t{i}a", f"t{i}b"]) for i in range(3)))
    outs = asyncio.run(go2())
    assert [len(o) for o in outs] == [2, 2, 2] and all(len(c) <= 3 for c in p2._model.calls) and p2.submissions == 3


def test_failure_reaches_every_caller_as_embedding_error():
    p = make({"batch_window_ms": 1})

    async def go():
        return await asyncio.gather(p._embed_batch_internal(["fine"]), p._embed_batch_internal(["poison"]), return_exceptions=True)
    res = asyncio.run(go())
    assert all(isinstance(r, EmbeddingError) and isinstance(r.cause, RuntimeError) for r in res)
    with pytest.raises(P.RetryError):                                                  # embed() retries 5x, then RetryError
        asyncio.run(p.embed("poison"))


def test_constructor_and_factory_surface(monkeypatch):
    p = P.HipUniXcoderProvider()
    assert p.config.provider == "unixcoder-hip" and p.config.model == "microsoft/unixcoder-base"
    assert p.embedding_dim == 768 and p.max_length == 512 and P.UniXcoderEmbeddingProvider is P.HipUniXcoderProvider
    monkeypatch.setenv("EMBEDDING_PROVIDER", "unixcoder")
    monkeypatch.setenv("CODERAG_HIP_WEIGHTS", "/models/unixcoder-base")
    q = P.get_embedding_provider()
    assert isinstance(q, P.HipUniXcoderProvider) and q.config.model == "/models/unixcoder-base"
    assert P.get_embedding_provider(provider="unixcoder-hip", model="/x").config.model == "/x"
    for name in ("openai", "anthropic", "nonsense"):
        with pytest.raises(ConfigurationError):
            P.get_embedding_provider(provider=name)
    cfg = P.ProviderConfig.from_env_prefix("NOPE")
    assert (cfg.provider, cfg.model) == ("openai", "gpt-4o")


def test_missing_checkpoint_is_a_loud_embedding_error():
    p = P.HipUniXcoderProvider(P.ProviderConfig(provider="unixcoder-hip", model="microsoft/unixcoder-base", extra={"dynamic_batching": False}))
    with pytest.raises(EmbeddingError) as e:
        asyncio.run(p._embed_batch_internal(["def f(): pass"]))
    assert "local" in str(e.value.cause) or "libcoderag" in str(e.value.cause) or "HIP" in str(e.value.cause)
