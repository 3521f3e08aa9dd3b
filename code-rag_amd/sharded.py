"""Row-sharded search across the GPUs of one node (no counterpart in the reference, whose
Qdrant server is a single process).

One process per GPU (``torch.distributed``; backend ``nccl`` = RCCL over xGMI on ROCm).  Every rank owns one
``crh_index`` shard; a search scans the local shard (``crh_search`` with ``row_base`` = the shard's global offset),
all-gathers the per-rank ``[nq, k]`` score / row lists -- the only exchange step of the path, 76.8 KB per rank at
nq=64, k=100, latency-bound -- and merges them on every rank (``crh_merge_topk``), so all ranks return the same
global top-k.  The embedding side needs no collective: each rank embeds its own chunks into its own shard.

The local index and the merge are injected (defaults: the HIP index and kernel) so the distributed logic --
offsets, all-gather layout, merge order, padding -- is exercised by world_size-2 ``gloo`` tests on CPU with the
oracle standing in for the device (tests/test_sharded_gloo.py).
"""

from __future__ import annotations

from typing import Any, Callable

import numpy as np


class ShardedIndex:
    def __init__(self, dim: int = 768, dtype: int | None = None, shard_capacity: int = 1 << 20, device: int = 0, group=None,
                 index_factory: Callable[..., Any] | None = None, merge_fn: Callable[..., Any] | None = None):
        import torch
        import torch.distributed as dist
        self._torch, self._dist, self.group = torch, dist, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.shard_capacity = int(shard_capacity)
        self.row_base = self.rank * self.shard_capacity            # global id of this shard's row 0: stable under appends
        self._on_device = index_factory is None
        if index_factory is None:
            from . import ffi
            dtype = ffi.DTYPE_BF16 if dtype is None else dtype
            self.index = ffi.Index(dim, dtype, capacity_rows=shard_capacity, device=device)
            self._merge = ffi.merge_topk
            self.device = torch.device("cuda", device)
        else:
            self.index = index_factory(dim=dim, dtype=dtype or 0, capacity_rows=shard_capacity, device=device)
            self._merge = merge_fn
            self.device = torch.device("cpu")
        self.dim = dim

    # ------------------------------------------------------------------ build
    def append_local(self, vecs, codes=None) -> tuple[int, int]:
        """Add rows to THIS rank's shard.  Returns (first global row, count)."""
        rows, _ = self.index.count()
        n = int(vecs.shape[0])
        if rows + n > self.shard_capacity:
            raise ValueError(f"shard of rank {self.rank} is full ({rows}+{n} > {self.shard_capacity})")
        first = self.index.append(vecs, codes)
        return self.row_base + first, n

    def append_scattered(self, vecs: np.ndarray, block: int = 4096) -> None:
        """Every rank is handed the SAME full array; blocks of `block` rows are dealt round-robin so shards stay
        balanced under incremental upserts.  Row r of the input lands on rank (r // block) % world."""
        for b0 in range(self.rank * block, len(vecs), self.world * block):
            self.append_local(vecs[b0:b0 + block])

    def global_counts(self) -> list[int]:
        torch, dist = self._torch, self._dist
        mine = torch.tensor([self.index.count()[0]], dtype=torch.int64, device=self.device)
        if self.world == 1:
            return [int(mine.item())]
        allc = torch.empty((self.world,), dtype=torch.int64, device=self.device)
        dist.all_gather_into_tensor(allc, mine, group=self.group)
        return [int(v) for v in allc.tolist()]

    # ------------------------------------------------------------------ query
    def search(self, queries, k: int, filters=None):
        """queries: [nq, dim] (replicated on every rank: numpy on the CPU test path, CUDA tensor or numpy on the
        device path).  Returns (scores [nq,k], rows [nq,k]) -- identical on every rank; rows are global ids."""
        torch, dist = self._torch, self._dist
        nq = int(queries.shape[0])
        # one record per rank, [scores | rows], so that ONE all-gather moves both (ffi.topk_exchange_buffers: plain torch views)
        from . import ffi
        local, loc_s, loc_r, gathered, all_s, all_r = ffi.topk_exchange_buffers(torch, self.world, nq, k, self.device)
        if self._on_device:
            ffi.use_device(self.device.index)
            stream = torch.cuda.current_stream(self.device).cuda_stream
            qd = queries if torch.is_tensor(queries) else torch.from_numpy(np.ascontiguousarray(queries, dtype=np.float32)).to(self.device)
            self.index.search(qd, k, filters=filters, row_base=self.row_base, out_scores=loc_s, out_rows=loc_r, stream=stream)
            self.index.search_finish(stream)
        else:
            s, r = self.index.search(np.asarray(queries, dtype=np.float32), k, filters=filters, row_base=self.row_base)
            loc_s.copy_(torch.from_numpy(np.ascontiguousarray(s)))
            loc_r.copy_(torch.from_numpy(np.ascontiguousarray(r)))
        if self.world == 1:
            return loc_s, loc_r
        # concatenated-along-dim-0 output form: the one every backend (RCCL and gloo) accepts
        dist.all_gather_into_tensor(gathered.view(-1), local, group=self.group)
        out_s = torch.empty((nq, k), dtype=torch.float32, device=self.device)
        out_r = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        if self._on_device:
            self._merge(all_s, all_r, out_s, out_r, torch.cuda.current_stream(self.device).cuda_stream)
        else:
            ms, mr = self._merge(all_s.numpy(), all_r.numpy())
            out_s.copy_(torch.from_numpy(ms))
            out_r.copy_(torch.from_numpy(mr))
        return out_s, out_r

    # ------------------------------------------------------------------ hybrid re-rank (BASELINE config 5)
    def attach_side_columns(self, side) -> None:
        """``side``: this shard's per-row side data (``ranking.device.SideColumns`` or anything with the same
        ``gather(rows, row_base=...)``).  The dictionary codes in it (file / merge key / centrality key) must come from
        dictionaries shared by all shards, since candidates of different shards are compared by code."""
        self.side = side

    def gather_columns(self, rows) -> dict[str, Any]:
        """Side data of a merged candidate table (global rows, identical on every rank): every rank gathers the rows it
        owns (the others give zeros) and ONE all-reduce(sum) of the packed buffer completes the table on all ranks -- [nq, k]
        small integers plus 64 name bytes per candidate, ~0.6 MB at nq=64, k=100 (one all-reduce per column for a plain dict)."""
        cols = self.side.gather(rows, row_base=self.row_base)
        if self.world > 1:
            packed = getattr(cols, "packed", None)
            if packed is not None:      # every column is a view of one buffer (ranking.device.PackedColumns): one collective
                self._dist.all_reduce(packed, op=self._dist.ReduceOp.SUM, group=self.group)
            else:
                for name in cols:
                    self._dist.all_reduce(cols[name], op=self._dist.ReduceOp.SUM, group=self.group)
        return cols

    def search_rerank(self, queries, k: int, plans, reranker, filters=None):
        """Global top-k of every query, then the hybrid re-rank of the merged lists (identical on every rank).
        Returns (RerankOutput, merged scores, merged global rows)."""
        s, r = self.search(queries, k, filters=filters)
        return reranker.rank(s, r, self.gather_columns(r), plans), s, r

    def owner_of(self, global_row: int) -> tuple[int, int]:
        """global row id -> (rank, local row)."""
        return int(global_row) // self.shard_capacity, int(global_row) % self.shard_capacity

    def close(self) -> None:
        self.index.close()
