"""The N>1 path on CPU: world_size-2 and -4 gloo processes, oracle-backed local shards, real all-gather, oracle merge.
The merged result on EVERY rank must equal the oracle's search over the concatenated corpus (ids mapped to the
shard offsets), including ragged shards, k larger than a shard, and filtered-out shards."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, out_dir: str) -> None:
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import coderag_amd  # noqa: F401
    from coderag_amd.sharded import ShardedIndex
    from oracle import search as orc
    from tests.fake_index import FakeIndex

    rng = np.random.default_rng(77)                         # same stream on every rank: replicated inputs
    x = rng.standard_normal((1000, 768)).astype(np.float32)
    q = rng.standard_normal((5, 768)).astype(np.float32)
    cap = 4096
    sh = ShardedIndex(768, 0, shard_capacity=cap, index_factory=FakeIndex, merge_fn=orc.merge_topk)
    sh.append_scattered(x, block=96)                        # 1000 rows in blocks of 96: ragged, unequal shards
    counts = sh.global_counts()
    assert sum(counts) == 1000 and len(set(counts)) > 1 and len(counts) == world

    # where did each input row go?  (block b -> rank b % world, appended in order)
    gid = np.empty(1000, dtype=np.int64)
    nxt = [0] * world
    for b0 in range(0, 1000, 96):
        r = (b0 // 96) % world
        n = min(96, 1000 - b0)
        gid[b0:b0 + n] = r * cap + nxt[r] + np.arange(n)
        nxt[r] += n

    for k in (10, 700):                                     # 700 > the smaller shard: padded local lists are merged correctly
        s, rows = sh.search(q, k)
        es, er = orc.cosine_search(x, q, k)
        exp_rows = np.where(er >= 0, gid[np.clip(er, 0, None)], -1)
        # ties are broken by lower GLOBAL id in the sharded run and by lower input row in the flat oracle: compare as sets per score
        assert np.array_equal(s.numpy(), es), f"rank {rank}: merged scores differ (k={k})"
        assert [sorted(a) for a in rows.numpy().tolist()] == [sorted(b) for b in exp_rows.tolist()], f"rank {rank}: ids differ (k={k})"
        gathered = [torch.empty_like(rows) for _ in range(world)]
        dist.all_gather(gathered, rows)
        assert all(torch.equal(gathered[0], g) for g in gathered), "ranks disagree on the merged result"
    assert sh.owner_of(int(rows[0, 0])) == (int(rows[0, 0]) // cap, int(rows[0, 0]) % cap)

    # side columns of the merged list (config 5): each rank contributes the rows it owns, one all-reduce completes them
    class FakeSide:
        def __init__(self, n_local):
            self.n = n_local

        def gather(self, rows_t, row_base=0):
            r = rows_t.numpy().reshape(-1) - row_base
            own = (rows_t.numpy().reshape(-1) >= 0) & (r >= 0) & (r < self.n)
            glob = rows_t.numpy().reshape(-1)
            val = np.where(own, (glob % 1000) * 3 + 1, 0).astype(np.int32)          # a function of the GLOBAL row id
            name = np.where(own[:, None], (glob[:, None] + np.arange(4)[None, :]) % 251, 0).astype(np.uint8)
            if not self.packed:
                return {"content_len": torch.from_numpy(val), "name": torch.from_numpy(name)}
            # the device form (ranking.device.PackedColumns): every column a view of one int32 buffer, one all-reduce
            from coderag_amd.ranking.device import PackedColumns
            n = len(val)
            buf = torch.zeros((2 * n,), dtype=torch.int32)
            out = PackedColumns()
            out.packed = buf
            out["content_len"] = buf[:n]
            out["name"] = buf[n:].view(torch.uint8).view(n, 4)
            out["content_len"].copy_(torch.from_numpy(val))
            out["name"].copy_(torch.from_numpy(name))
            return out

    for packed in (False, True):
        side = FakeSide(counts[rank])
        side.packed = packed
        sh.attach_side_columns(side)
        s, rows = sh.search(q, 10)
        cols = sh.gather_columns(rows)
        flat = rows.numpy().reshape(-1)
        assert np.array_equal(cols["content_len"].numpy(), ((flat % 1000) * 3 + 1).astype(np.int32)), f"rank {rank}: column not completed"
        assert np.array_equal(cols["name"].numpy(), ((flat[:, None] + np.arange(4)[None, :]) % 251).astype(np.uint8))
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])      # 4: shards of 288 / 288 / 232 / 192 rows, k=700 larger than every one of them
def test_world_size_n_gloo(tmp_path, world):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == [f"ok{r}" for r in range(world)]
