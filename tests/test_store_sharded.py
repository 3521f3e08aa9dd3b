"""The row-sharded index BEHIND the reference's store surface (BASELINE configs[3]; north_star: "keeping the existing ...
plugin surfaces ... the corpus shards row-wise across the GPUs"): ``HipVectorStore(shards=N)`` driven by the same scenarios as
the unsharded store -- in-process shards, and one process per shard under torch.distributed (gloo, world 2) -- with the
oracle-backed FakeIndex standing in for the device.  Results must equal ``shards=1``.  GPU tier: tests/test_store_gpu.py.
Reference callers this serves: embeddings/client.py:115-169, embeddings/indexer.py:46-94, query/vector_search.py:60-116."""
import asyncio
import os
import socket
import sys
import uuid

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _patch_fake():
    import coderag_amd  # noqa: F401
    from coderag_amd import ffi
    from tests.fake_index import FakeIndex
    ffi.Index = FakeIndex
    ffi.lib = lambda: object()
    ffi.device_count = lambda: 1
    ffi.device_info = lambda d=0: {"name": "fake", "arch": "gfx950", "hbm_bytes": 0, "cu_count": 256}


@pytest.fixture
def fake(monkeypatch):
    import coderag_amd  # noqa: F401
    from coderag_amd import ffi
    from tests.fake_index import FakeIndex
    monkeypatch.setattr(ffi, "Index", FakeIndex)
    monkeypatch.setattr(ffi, "lib", lambda: object())
    monkeypatch.setattr(ffi, "device_count", lambda: 1)
    monkeypatch.setattr(ffi, "device_info", lambda d=0: {"name": "fake", "arch": "gfx950", "hbm_bytes": 0, "cu_count": 256})


def _store(shards, **kw):
    from coderag_amd import store as store_mod
    from oracle import search as orc
    return store_mod.HipVectorStore(dim=768, initial_capacity=64, shards=shards, _merge_fn=orc.merge_topk, **kw)


async def drive(s, log, snap_dir=None):
    """A life of a collection: bulk upsert (blocks over the shards), searches, filters, replace, delete by file, re-index of
    the same files (the reference's flow, embeddings/indexer.py:61-85), compaction, snapshot.  Every observable result is
    appended to ``log`` so that two stores can be compared."""
    rng = np.random.default_rng(9)
    n = 900
    vecs = rng.standard_normal((n, 768)).astype(np.float32)
    ids = [str(uuid.UUID(int=int(v))) for v in rng.integers(1, 1 << 62, n)]
    pay = [{"file_path": f"/p/f{i % 30}.py", "entity_type": "function", "entity_name": f"e{i}", "language": ("python", "go", "rust")[i % 3],
            "start_line": i, "end_line": i + 2, "content": f"body {i}", "graph_node_id": None, "content_hash": f"h{i % 30}", "project_name": "p"} for i in range(n)]
    qs = rng.standard_normal((6, 768)).astype(np.float32)

    def norm(hits):
        return [(h["id"], h["score"], h["payload"]) for h in hits]
    async with s:
        await s.create_collections()
        for a in range(0, n, 250):                                  # blocks of 4096 rows per shard turn: use a smaller block below
            await s.upsert("code_chunks", ids[a:a + 250], vecs[a:a + 250], pay[a:a + 250])
        log.append((await s.get_collection_info("code_chunks")).points_count)
        for q in qs:
            log.append(norm(await s.search("code_chunks", q.tolist(), limit=15)))
            log.append(norm(await s.search("code_chunks", q.tolist(), limit=40, filters={"language": "go"})))
        log.append([norm(h) for h in await s.search_batch("code_chunks", qs, limit=9, filters={"language": "rust"})])
        log.append(norm(await s.search("code_chunks", None, limit=7, filters={"file_path": "/p/f3.py"})))
        log.append([await s.file_needs_update("code_chunks", f"/p/f{i}.py", f"h{i}") for i in (0, 5, 29)] + [await s.file_needs_update("code_chunks", "/p/f1.py", "other")])
        # the reference's re-index of a file: delete by file_path, then upsert its chunks again under fresh ids -- five files, three rounds
        for rnd in range(3):
            for f in range(5):
                await s.delete("code_chunks", {"file_path": f"/p/f{f}.py"})
                rows = [i for i in range(n) if i % 30 == f]
                await s.upsert("code_chunks", [str(uuid.UUID(int=1000 * rnd + i + 1)) for i in rows], vecs[rows] + 0.001 * (rnd + 1),
                               [dict(pay[i], content_hash=f"r{rnd}") for i in rows])
        log.append((await s.get_collection_info("code_chunks")).points_count)
        log.append(norm(await s.search("code_chunks", qs[0].tolist(), limit=30)))
        # replace by id, delete through the raw client (MatchText), counts
        await s.upsert("code_chunks", [ids[100], ids[101]], -vecs[100:102], [dict(pay[100], content="changed"), pay[101]])
        log.append(norm(await s.search("code_chunks", (-vecs[100]).tolist(), limit=2)))
        from types import SimpleNamespace as M
        flt = M(must=[M(key="file_path", match=M(text="/p/f2"))])
        log.append((await s.client.count("code_chunks", count_filter=flt)).count)
        await s.client.delete("code_chunks", points_selector=M(filter=flt))
        log.append((await s.get_collection_info("code_chunks")).points_count)
        before = norm(await s.search("code_chunks", qs[1].tolist(), limit=25))
        reclaimed = await s.compact("code_chunks")
        info = await s.get_collection_info("code_chunks")
        assert reclaimed >= 0 and info.config["rows_appended"] == info.points_count and info.config["compactions"] >= 1
        assert norm(await s.search("code_chunks", qs[1].tolist(), limit=25)) == before          # compaction is invisible
        log.append(before)
        log.append(norm(await s.search("code_chunks", None, limit=50, filters={"file_path": "/p/f7.py"})))
        await s.upsert("code_chunks", [ids[7]], vecs[7:8], [dict(pay[7], language="zig")])        # tables are live after compaction
        log.append(norm(await s.search("code_chunks", vecs[7].tolist(), limit=3, filters={"language": "zig"})))
        import tempfile
        with tempfile.TemporaryDirectory() as snap:
            snap = snap_dir or snap                                   # (ranks of one store share the snapshot directory)
            await s.save(snap)
            await s.load(snap)
        log.append(norm(await s.search("code_chunks", qs[2].tolist(), limit=25)))
        log.append([await s.file_needs_update("code_chunks", "/p/f0.py", "r2"), await s.file_needs_update("code_chunks", "/p/f2.py", "h2")])
        log.append((await s.get_collection_info("code_chunks")).points_count)


def test_store_scenarios_with_three_in_process_shards(fake):
    """tests/store_scenarios.py (incl. everything the unsharded CPU tier runs) on three shards."""
    from tests.store_scenarios import run_store_scenarios
    asyncio.run(run_store_scenarios(_store(3)))


def test_reference_database_scenario_with_three_shards(fake, monkeypatch):
    from coderag_amd import store as store_mod
    from tests.store_scenarios import run_reference_database_scenario
    monkeypatch.delenv("EMBEDDING_DIMENSIONS", raising=False)
    monkeypatch.delenv("EMBEDDING_PROVIDER", raising=False)
    monkeypatch.setenv("CODERAG_HIP_SHARDS", "3")
    from oracle import search as orc
    asyncio.run(run_reference_database_scenario(store_mod.QdrantManager(_merge_fn=orc.merge_topk), store_mod.CollectionName))


def test_sharded_store_equals_the_unsharded_one(fake, monkeypatch):
    """The same life of a collection on 1, 2 and 3 shards (blocks of 64 rows per shard turn): every search, count, update check
    and filter-only fetch gives the same ids, scores and payloads; the shards are balanced; auto-compaction ran."""
    from coderag_amd import shards as shards_mod
    logs = {}
    for ns in (1, 2, 3):
        s = _store(ns, compact_dead_fraction=0.2, compact_min_dead=50)
        orig = shards_mod.ShardSet.__init__

        def small_blocks(self, *a, _orig=orig, **kw):
            kw["block"] = 64
            _orig(self, *a, **kw)
        monkeypatch.setattr(shards_mod.ShardSet, "__init__", small_blocks)
        logs[ns] = []
        asyncio.run(drive(s, logs[ns]))
        monkeypatch.setattr(shards_mod.ShardSet, "__init__", orig)
    assert logs[1] == logs[2] == logs[3]
    assert len(logs[1]) > 25


def _dist_worker(rank: int, world: int, port: int, out_dir: str) -> None:
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _patch_fake()
    from coderag_amd import shards as shards_mod
    orig = shards_mod.ShardSet.__init__

    def small_blocks(self, *a, **kw):
        kw["block"] = 64
        orig(self, *a, **kw)
    shards_mod.ShardSet.__init__ = small_blocks
    # every rank makes the same calls (replicated host tables, sharded vectors): backend "dist" is picked up from the process group
    s = _store(world, compact_dead_fraction=0.2, compact_min_dead=50)
    assert s._shard_backend == "dist"
    log = []
    asyncio.run(drive(s, log, snap_dir=os.path.join(out_dir, "snap")))
    shards_mod.ShardSet.__init__ = orig
    ref = []
    asyncio.run(drive(_store(1, compact_dead_fraction=0.2, compact_min_dead=50), ref))       # the unsharded store, in this very process
    assert log == ref, f"rank {rank}: the sharded store and the unsharded one disagree"
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    dist.destroy_process_group()


def test_store_on_one_process_per_shard_gloo_world_2(tmp_path):
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    mp.spawn(_dist_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert {"ok0", "ok1"} <= set(os.listdir(tmp_path))
