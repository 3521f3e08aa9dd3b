"""Row shards of one collection behind ``HipVectorStore`` (BASELINE configs[3] / [4]: "corpus row-sharded across the GPUs").

``north_star`` keeps the reference's ``VectorStore`` surface (``embeddings/client.py:18-228``) and shards the corpus behind it;
this is the piece between the two.  A ``ShardSet`` looks like one index to ``store._Collection`` -- append, tombstone,
delete / count / match by filter, search, compact, save / load -- and spreads the rows over ``n`` ``crh_index`` handles:

* ``backend="local"``: all shards live in this process (on one device, or one per listed device).  What a one-GPU box can run;
  the search is n scans -> ``crh_merge_topk_strided`` on the device.
* ``backend="dist"``: one process per GPU under ``torch.distributed`` (``nccl`` = RCCL over xGMI); every rank makes the SAME
  calls and owns shard ``rank``: its rows' vectors, and -- since round 4 -- their embedding work and their payload TEXT
  (``append`` takes the rows of the owned shards only; ``exchange_bytes`` completes what only an owner holds).  Ids, the coded
  payload columns and the slot maps stay replicated.  The search is the path of ``sharded.ShardedIndex``: local scan with
  ``row_base``, ONE all-gather of the ``[scores | rows]`` records, merge on every rank.  Counts, matching rows and compaction maps travel in tensor
  collectives as well (round 4; a compaction map is one entry per row).  An append's outcome is agreed in ONE one-integer all-reduce;
  only a FAILED append exchanges small host objects (the reason, and the rows that landed before the failure).

Rows are dealt to the shards in blocks of ``block`` rows, round robin, so shards stay balanced under incremental upserts.
A row's global id is ``shard * STRIDE + local row`` (``STRIDE`` = 2^32, a ``crh_index`` holds at most 2^31 rows): stable under
appends and capacity growth.  Ties between equal scores go to the lower global id (lower shard first) -- Qdrant leaves tie
order unspecified.
"""

from __future__ import annotations

import os
from typing import Any, Callable, Sequence

import numpy as np

from . import ffi

STRIDE = 1 << 32


class AppendFailed(RuntimeError):
    """An append that failed after some shards had taken their rows: ``done`` = {shard: (first local row, rows)} of the rows
    that ARE on the device (tombstoned by the time this is raised); the caller's slot maps must step over them."""

    def __init__(self, cause: BaseException, done: dict[int, tuple[int, int]]):
        super().__init__(f"append failed: {cause!r}")
        self.cause, self.done = cause, done


class ShardSet:
    def __init__(self, nshards: int, make_index: Callable[[int], Any], device: int = 0, backend: str = "local", group=None,
                 block: int = 4096, merge_fn: Callable | None = None):
        if nshards < 1:
            raise ValueError("a collection needs at least one shard")
        self.ns, self.block, self.device, self._merge_host = int(nshards), int(block), device, merge_fn
        self.backend, self.group, self.dist, self.rank = backend, group, None, None
        if backend == "dist":
            import torch.distributed as dist
            if not dist.is_initialized():
                raise RuntimeError("shard backend 'dist' needs an initialised torch.distributed process group (one process per GPU)")
            if dist.get_world_size(group) != self.ns:
                raise ValueError(f"{self.ns} shards but a process group of {dist.get_world_size(group)} ranks")
            self.dist, self.rank = dist, dist.get_rank(group)
            self.owned = [self.rank]
        elif backend == "local":
            self.owned = list(range(self.ns))
        else:
            raise ValueError(f"unknown shard backend {backend!r} (use 'local' or 'dist')")
        self.index = {s: make_index(s) for s in self.owned}
        self.stream = 0                           # raw hipStream_t of the stream searches run on (0: the default stream); the store sets it
        self.rows = [0] * self.ns                 # rows appended to every shard so far (replicated bookkeeping: same on every rank)
        self._next_block = 0
        first = self.index[self.owned[0]]
        self.dim, self.dtype = first.dim, first.dtype

    # ------------------------------------------------------------------ plumbing
    def _everyone(self, mine: Any) -> list:
        """``mine`` of every rank, in rank order (backend "dist"); a one-element list otherwise."""
        if self.dist is None:
            return [mine]
        out = [None] * self.ns
        self.dist.all_gather_object(out, mine, group=self.group)
        return out

    def _tensor_device(self):
        import torch
        return torch.device("cuda", self.device) if self.dist.get_backend(self.group) == "nccl" else torch.device("cpu")

    def _sum_everyone(self, value: int) -> int:
        """Sum of one integer per rank: ONE tensor all-reduce (counts of deleted / matching / alive rows)."""
        if self.dist is None:
            return int(value)
        import torch
        t = torch.tensor([int(value)], dtype=torch.int64, device=self._tensor_device())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return int(t.item())

    def _arrays_everyone(self, mine: dict[int, np.ndarray]) -> dict[int, np.ndarray]:
        """{shard: int64 array} of every rank, united -- two tensor collectives (lengths, then the padded arrays) instead of
        pickled objects: compaction maps are one entry per ROW (80 MB per 10M-row shard)."""
        if self.dist is None:
            return {s: np.asarray(a, np.int64) for s, a in mine.items()}
        import torch
        dev = self._tensor_device()
        arr = np.asarray(mine[self.rank], np.int64) if self.rank in mine else np.zeros((0,), np.int64)
        lens = torch.zeros((self.ns,), dtype=torch.int64, device=dev)
        self.dist.all_gather_into_tensor(lens, torch.tensor([arr.size], dtype=torch.int64, device=dev), group=self.group)
        lens_h = lens.cpu().numpy()
        width = max(1, int(lens_h.max()))
        pad = np.zeros((width,), np.int64)
        pad[:arr.size] = arr
        out = torch.empty((self.ns, width), dtype=torch.int64, device=dev)
        self.dist.all_gather_into_tensor(out.view(-1), torch.from_numpy(pad).to(dev), group=self.group)
        out_h = out.cpu().numpy()
        return {r: out_h[r, : int(lens_h[r])].copy() for r in range(self.ns)}

    def barrier(self) -> None:
        if self.dist is not None:
            self.dist.barrier(group=self.group)

    @property
    def capacity_rows(self) -> int:
        return sum(ix.capacity_rows for ix in self.index.values())

    def count(self) -> tuple[int, int]:
        alive = self._sum_everyone(sum(ix.count()[1] for ix in self.index.values()))
        return sum(self.rows), int(alive)

    # ------------------------------------------------------------------ build
    def route(self, n: int) -> np.ndarray:
        """Shard of each of the next ``n`` appended rows: blocks of ``block`` rows, round robin, continuing where the previous
        append stopped."""
        if self.ns == 1:
            return np.zeros((n,), np.int32)
        blocks = (n + self.block - 1) // self.block
        sh = (np.arange(blocks, dtype=np.int64) + self._next_block) % self.ns
        self._next_block = int((self._next_block + blocks) % self.ns)
        return np.repeat(sh, self.block)[:n].astype(np.int32)

    def append(self, vecs, codes, preprocessed: bool = False, stream: int = 0, shard: np.ndarray | None = None,
               failure: BaseException | None = None) -> tuple[np.ndarray, np.ndarray]:
        """Append n rows.  ``vecs``: numpy [n, dim] or a CUDA tensor holding ALL n rows -- or, with ``shard`` (the routing of the
        n rows, from :meth:`route`), a dict {owned shard: its rows in order}: the form a rank uses when it has embedded only its
        own share.  ``codes``: numpy [n, cols] for all n rows, or None.  Returns (shard [n], local row [n]).

        Nothing is committed until every owned shard has the capacity; if a shard's append still fails, the rows the earlier
        shards took are tombstoned and :class:`AppendFailed` tells the caller where they sit.  Under backend "dist" the ranks
        agree on the outcome before anyone returns (one one-integer all-reduce when all is well).

        ``failure``: an error this process met while PREPARING its rows (its share of the texts would not embed): nothing is
        appended here, but the call is still made so that every rank takes part in the same agreement and leaves the same way --
        a rank that raised before this point would leave the others waiting in the collective."""
        per_shard = isinstance(vecs, dict)
        if per_shard and shard is None:
            raise ValueError("per-shard rows need the routing they were cut by")
        n = int(len(shard) if shard is not None else vecs.shape[0])
        saved_next = self._next_block
        if shard is None:
            shard = self.route(n)
        local = np.empty((n,), np.int64)
        plan = []                                     # (shard, rows of the call, first local row)
        for s in range(self.ns):
            sel = np.flatnonzero(shard == s) if self.ns > 1 else np.arange(n)
            if sel.size:
                local[sel] = self.rows[s] + np.arange(sel.size)
                plan.append((s, sel, self.rows[s]))
        done: dict[int, tuple[int, int]] = {}
        try:
            if failure is not None:
                raise failure
            for s, sel, first in plan:                # capacity first, on every owned shard: a refusal here leaves nothing behind
                if s in self.index and first + sel.size > self.index[s].capacity_rows:
                    self.index[s].reserve(max(first + int(sel.size), 2 * self.index[s].capacity_rows))
            for s, sel, first in plan:
                if s not in self.index:
                    continue
                ix = self.index[s]
                if per_shard:
                    v = vecs[s]
                elif self.ns == 1:
                    v = vecs
                elif isinstance(vecs, np.ndarray):
                    v = vecs[sel]
                else:
                    import torch
                    v = vecs.index_select(0, torch.from_numpy(sel).to(vecs.device))
                if int(v.shape[0]) != sel.size:
                    raise ValueError(f"shard {s}: {int(v.shape[0])} rows for {sel.size} routed to it")
                c = None if codes is None else (codes if self.ns == 1 else codes[sel])
                on_dev = not isinstance(v, np.ndarray)
                if on_dev and c is not None:
                    import torch
                    c = torch.from_numpy(np.ascontiguousarray(c)).to(v.device)
                got = ix.append(v, c, stream=stream, preprocessed=preprocessed) if on_dev else ix.append(v, c, preprocessed=preprocessed)
                done[s] = (int(got), int(sel.size))
                if got != first:
                    raise RuntimeError(f"shard {s}: append landed at row {got}, the bookkeeping expected {first}")
        except BaseException as e:  # noqa: BLE001 -- rolled back below, then re-raised
            failure = e
        if self.dist is not None:                     # one rank's failure is everybody's: the replicated bookkeeping must not part ways
            if self._sum_everyone(0 if failure is None else 1):
                outcomes = self._everyone(None if failure is None else repr(failure))
                if failure is None:
                    failure = RuntimeError(f"append failed on another rank: {[o for o in outcomes if o is not None][0]}")
        if failure is not None:
            for s, (got, m) in done.items():          # the rows that did land: dead, and accounted for
                self.index[s].tombstone(np.arange(got, got + m, dtype=np.int64))
            all_done: dict[int, tuple[int, int]] = {}
            for part in self._everyone(done):
                all_done.update(part)
            for s, (got, m) in all_done.items():
                self.rows[s] = got + m
            self._next_block = saved_next if not all_done else self._next_block
            if all_done:
                raise AppendFailed(failure, all_done) from failure
            raise failure
        for s, sel, first in plan:
            self.rows[s] = first + int(sel.size)
        return shard, local

    def exchange_bytes(self, parts: list) -> list:
        """``parts[i]``: bytes where THIS rank holds item i, None elsewhere (every item is held by exactly one rank, and every rank
        passes a list of the same length).  Returns the complete list on every rank -- two tensor all-reduces (lengths, then one
        byte buffer), no pickled objects: what a search's hits need from the ranks that own their payload text."""
        if self.dist is None:
            return parts
        import torch
        dev = torch.device("cuda", self.device) if self.dist.get_backend(self.group) == "nccl" else torch.device("cpu")
        lens = torch.tensor([0 if p is None else len(p) for p in parts], dtype=torch.int64, device=dev)
        self.dist.all_reduce(lens, op=self.dist.ReduceOp.SUM, group=self.group)
        lens_h = lens.cpu().numpy()
        off = np.zeros((len(parts) + 1,), np.int64)
        np.cumsum(lens_h, out=off[1:])
        buf = np.zeros((max(int(off[-1]), 1),), np.uint8)
        for i, p in enumerate(parts):
            if p:
                if len(p) != int(lens_h[i]):
                    raise RuntimeError("exchange_bytes: an item is held by more than one rank")
                buf[off[i]:off[i + 1]] = np.frombuffer(p, np.uint8)
        t = torch.from_numpy(buf).to(dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        out_b = t.cpu().numpy().tobytes()
        return [out_b[off[i]:off[i + 1]] for i in range(len(parts))]

    def tombstone(self, shard: np.ndarray, local: np.ndarray) -> None:
        for s, ix in self.index.items():
            sel = local[shard == s] if self.ns > 1 else local
            if len(sel):
                ix.tombstone(np.asarray(sel, np.int64))

    def tombstone_filter(self, dfilt) -> int:
        return self._sum_everyone(sum(ix.tombstone_filter(dfilt) for ix in self.index.values()))

    def count_matching(self, dfilt) -> int:
        return self._sum_everyone(sum(ix.count_matching(dfilt) for ix in self.index.values()))

    def match_rows(self, dfilt, limit: int) -> tuple[np.ndarray, np.ndarray]:
        """(shard, local row) of up to ``limit`` alive matching rows PER SHARD, ascending inside each shard (a caller that wants
        the first ``limit`` in insertion order sorts the union by slot and cuts)."""
        allr = self._arrays_everyone({s: ix.match_rows(dfilt, limit) for s, ix in self.index.items()})
        sh = np.concatenate([np.full((len(allr[s]),), s, np.int32) for s in sorted(allr)]) if allr else np.zeros((0,), np.int32)
        lo = np.concatenate([np.asarray(allr[s], np.int64) for s in sorted(allr)]) if allr else np.zeros((0,), np.int64)
        return sh, lo

    # ------------------------------------------------------------------ query
    def search(self, queries: np.ndarray, k: int, dfilt) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
        """Exact top-k over all shards: (scores [nq, k], shard [nq, k], local row [nq, k]); -1 rows are padding."""
        if self.ns == 1:
            s, r = self.index[0].search(queries, k, filters=dfilt, **({"stream": self.stream} if self.stream else {}))
            return s, np.zeros(r.shape, np.int32), r
        if self._merge_host is not None:            # injected host-side index + merge (CPU test tier)
            scores, rows = self._search_host(queries, k, dfilt)
        else:
            sd, rd = self.search_device(queries, k, dfilt)
            scores, rows = sd.cpu().numpy(), rd.cpu().numpy()
        return scores, np.where(rows >= 0, rows // STRIDE, 0).astype(np.int32), np.where(rows >= 0, rows % STRIDE, -1)

    def _search_host(self, queries, k, dfilt):
        nq = int(queries.shape[0])
        mine = {s: ix.search(queries, k, filters=dfilt, row_base=s * STRIDE) for s, ix in self.index.items()}
        if self.dist is None:
            parts = mine
        else:                                        # the same single all-gather of [scores | rows] records as on the device
            import torch
            local, loc_s, loc_r, gathered, all_s, all_r = ffi.topk_exchange_buffers(torch, self.ns, nq, k, torch.device("cpu"))
            ms, mr = mine[self.rank]
            loc_s.copy_(torch.from_numpy(np.ascontiguousarray(ms)))
            loc_r.copy_(torch.from_numpy(np.ascontiguousarray(mr)))
            self.dist.all_gather_into_tensor(gathered.view(-1), local, group=self.group)
            return self._merge_host(all_s.numpy(), all_r.numpy())
        ss = np.stack([parts[s][0] for s in range(self.ns)])
        rr = np.stack([parts[s][1] for s in range(self.ns)])
        return self._merge_host(ss, rr)

    def search_device(self, queries, k: int, dfilt):
        """The same on the device, results left there: (scores f32 [nq, k], GLOBAL rows i64 [nq, k]) CUDA tensors -- per-shard
        ``crh_search`` with ``row_base`` = shard * STRIDE, [one all-gather of the records,] ``crh_merge_topk_strided``."""
        import torch
        dev = torch.device("cuda", self.device)
        ffi.use_device(self.device)
        stream = torch.cuda.current_stream(dev).cuda_stream
        qd = queries if torch.is_tensor(queries) else torch.from_numpy(np.ascontiguousarray(queries, dtype=np.float32)).to(dev)
        nq = int(qd.shape[0])
        local, loc_s, loc_r, gathered, all_s, all_r = ffi.topk_exchange_buffers(torch, self.ns, nq, k, dev)
        if self.dist is None:
            for s, ix in self.index.items():         # every shard writes its own record of the "gathered" buffer
                ix.search(qd, k, filters=dfilt, row_base=s * STRIDE, out_scores=all_s[s], out_rows=all_r[s], stream=stream)
            for ix in self.index.values():
                ix.search_finish(stream)
        else:
            ix = self.index[self.rank]
            ix.search(qd, k, filters=dfilt, row_base=self.rank * STRIDE, out_scores=loc_s, out_rows=loc_r, stream=stream)
            ix.search_finish(stream)
            self.dist.all_gather_into_tensor(gathered.view(-1), local, group=self.group)
        out_s = torch.empty((nq, k), dtype=torch.float32, device=dev)
        out_r = torch.empty((nq, k), dtype=torch.int64, device=dev)
        ffi.merge_topk(all_s, all_r, out_s, out_r, stream)
        return out_s, out_r

    def complete_columns(self, packed) -> None:
        """Side columns of a merged candidate table: every rank gathered the rows it owns (zeros elsewhere); ONE all-reduce of
        the packed buffer completes them (``sharded.ShardedIndex.gather_columns``).  Local shards are summed by the caller."""
        if self.dist is not None:
            self.dist.all_reduce(packed, op=self.dist.ReduceOp.SUM, group=self.group)

    # ------------------------------------------------------------------ maintenance
    def compact(self) -> dict[int, np.ndarray]:
        """``crh_index_compact`` on every shard; returns {shard: old_to_new local rows} for ALL shards on every rank."""
        maps = self._arrays_everyone({s: ix.compact() for s, ix in self.index.items()})
        for s, o2n in maps.items():
            self.rows[s] = int((o2n >= 0).sum())
        return maps

    def stats(self) -> dict:
        out: dict[str, int] = {}
        for ix in self.index.values():
            for k, v in ix.stats().items():
                out[k] = (max(out.get(k, 0), v) if k in ("max_query_cands", "fallback_used") else out.get(k, 0) + v)
        return out

    def save(self, directory: str) -> None:
        for s, ix in self.index.items():
            ix.save(directory if self.ns == 1 else os.path.join(directory, f"shard{s}"))

    def load(self, directory: str) -> None:
        for s, ix in self.index.items():
            ix.load(directory if self.ns == 1 else os.path.join(directory, f"shard{s}"))
        counts = self._arrays_everyone({s: np.asarray([ix.count()[0]], np.int64) for s, ix in self.index.items()})
        self.rows = [int(counts[s][0]) for s in range(self.ns)]

    def close(self) -> None:
        for ix in self.index.values():
            ix.close()


def shard_sizes(rows: Sequence[int]) -> str:
    return "/".join(str(int(r)) for r in rows)
