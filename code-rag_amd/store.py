"""HipVectorStore -- the reference's ``QdrantManager`` surface on an HBM-resident HIP index.

Drop-in for ``src/lattice/embeddings/client.py:18-228`` (and the ``VectorStore`` protocol of
``src/lattice/core/protocols.py:34-53``): same method names, argument meaning, return shapes and
error behaviour, but the vectors live in this process's GPU instead of a Qdrant server and the
cosine top-k runs in ``libcoderag_hip.so`` (``crh_search``).  Payload dictionaries, point ids and
the value<->code dictionaries of the filterable payload keys stay on the host; the device only
sees int32 codes.

There is no CPU fallback: without the native library or a gfx950 device ``connect()`` raises
``VectorStoreError``.
"""

from __future__ import annotations

import asyncio
import os
import logging
import threading
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass, field
from enum import Enum
from typing import Any

import numpy as np

from . import ffi
from .errors import VectorStoreError
from .settings import get_settings
from .shards import STRIDE as SHARD_STRIDE, AppendFailed, ShardSet
from .tables import DICT_KEYS, TEXT_KEYS, IdTable, PayloadTable

logger = logging.getLogger(__name__)


class CollectionName(str, Enum):
    """embeddings/client.py:13-15"""

    CODE_CHUNKS = "code_chunks"
    SUMMARIES = "summaries"


# Payload keys that can appear in a filter, per collection.  The first group mirrors the keyword payload
# indexes the reference creates (client.py:77-89); the rest are keys its callers filter on without an index
# (entity_name: query/context/builder.py:111-119; project_name on summaries: query/vector_search.py:144-146).
FILTER_KEYS: dict[str, tuple[str, ...]] = {
    CollectionName.CODE_CHUNKS.value: ("file_path", "entity_type", "language", "content_hash", "project_name",
                                       "entity_name"),
    CollectionName.SUMMARIES.value: ("file_path", "entity_type", "entity_name", "project_name"),
}

_DTYPES = {"f32": ffi.DTYPE_F32, "fp32": ffi.DTYPE_F32, "float32": ffi.DTYPE_F32, "bf16": ffi.DTYPE_BF16,
           "bfloat16": ffi.DTYPE_BF16}


@dataclass
class CollectionInfo:
    """What ``get_collection_info`` hands back; callers read ``points_count`` (query/engine.py:299-302)."""

    name: str
    points_count: int
    vectors_count: int
    indexed_vectors_count: int
    status: str = "green"
    config: dict = field(default_factory=dict)


class _Collection:
    """One collection: row shards on the device(s) + the host-side id / payload tables.

    What lives where: vectors, validity bits and the dictionary CODES of the filterable payload keys are on the device
    (columnar int32, one column per key; ``shards.ShardSet`` spreads the rows over one or more ``crh_index`` handles).  The
    host keeps what a hit has to carry back -- ids and payloads, by SLOT (insertion order), columnar (``tables.IdTable`` /
    ``tables.PayloadTable``: no Python object per row) -- and the map slot <-> (shard, local row).  Filters, deletes and the
    update check are resolved on the device from the code columns.  A deleted row keeps its slot until ``compact()``
    (``crh_index_compact``) moves the survivors together on the device and in the tables; the store compacts by itself once
    the dead rows pass a fraction of the collection (the reference deletes and re-inserts every chunk of a file on every
    indexing run, embeddings/indexer.py:61-64)."""

    def __init__(self, name: str, dim: int, dtype: int, capacity: int, device: int, nshards: int = 1, backend: str = "local",
                 group=None, merge_fn=None, compact_dead_fraction: float = 0.25, compact_min_dead: int = 1024):
        self.name = name
        self.keys = FILTER_KEYS.get(name, ())
        ncols = len(self.keys)
        self.shards = ShardSet(nshards, lambda s: ffi.Index(dim, dtype, capacity_rows=capacity, n_code_cols=ncols, device=device),
                               device=device, backend=backend, group=group, merge_fn=merge_fn)
        self.ids = IdTable()
        self.payloads = PayloadTable(tuple(DICT_KEYS) + tuple(self.keys))
        # slot <-> row.  With ONE shard a slot IS its row (rows are appended and compacted in slot order): no map is kept.
        self.row_shard = np.zeros((0,), np.int32)
        self.row_local = np.zeros((0,), np.int64)
        self.slot_of: list[np.ndarray] = [np.zeros((0,), np.int64) for _ in range(self.shards.ns)]
        self._side: dict[int, Any] = {}      # shard -> ranking.device.SideColumns of its rows (built on first use)
        self._side_books = None
        self._degrees: dict[str, int] | None = None
        self._device = device
        self.compact_dead_fraction, self.compact_min_dead = compact_dead_fraction, compact_min_dead
        self.compactions = 0
        # One process per shard (backend "dist"): a rank keeps the payload TEXT (content, summary: 4.2 of the 5.2 GB of host
        # tables per 10M chunks) of its OWN rows only -- everybody else stores an empty string there -- and a hit's payload comes
        # from the rank that owns the row (payloads_of: one byte exchange per search, shards.ShardSet.exchange_bytes).
        self.partial = backend == "dist" and nshards > 1

    @property
    def index(self):
        """The one ``crh_index`` of an unsharded collection (tools and tests look at it)."""
        if self.shards.ns != 1:
            raise AttributeError("a sharded collection has no single index (use .shards)")
        return self.shards.index[0]

    # -- slots and rows
    def rows_of(self, slots: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
        slots = np.asarray(slots, np.int64)
        if self.shards.ns == 1:
            return np.zeros(slots.shape, np.int32), slots
        return self.row_shard[slots], self.row_local[slots]

    def slots_of(self, shard: np.ndarray, local: np.ndarray) -> np.ndarray:
        """Slots of rows (shard, local); -1 stays -1."""
        local = np.asarray(local, np.int64)
        if self.shards.ns == 1:
            return local
        out = np.full(local.shape, -1, np.int64)
        for s in range(self.shards.ns):
            m = (np.asarray(shard) == s) & (local >= 0)
            if m.any():
                out[m] = self.slot_of[s][local[m]]
        return out

    def side_columns(self) -> dict[int, Any]:
        """Per-shard side data for the device re-rank (ranking/device.py), extended lazily as rows are appended; the
        dictionaries behind the file / merge-key / node codes are shared by the shards (candidates of different shards are
        compared by code)."""
        from .ranking.device import SideColumns
        if self._side_books is None:
            self._side_books = {"file": {}, "key": {}, "node": {}}
        for s in self.shards.owned:
            side = self._side.get(s)
            if side is None:
                side = self._side[s] = SideColumns(self._device, books=self._side_books)
            have, want = side.rows, self.shards.rows[s]
            if have < want:
                slots = np.arange(have, want) if self.shards.ns == 1 else self.slot_of[s][have:want]
                side.append([self.payloads.get(int(t)) if t >= 0 else {} for t in slots])      # (-1: a dead row without a slot)
                if self._degrees is not None:
                    side.set_degrees(self._degrees)
        return self._side

    def gather_side(self, rows_dev):
        """Side columns of a candidate table of GLOBAL rows (CUDA int64 [nq, k]), complete over all shards."""
        sides = self.side_columns()
        total = None
        for s in self.shards.owned:
            cols = sides[s].gather(rows_dev, row_base=s * SHARD_STRIDE)
            if total is None:
                total = cols
            else:
                total.packed += cols.packed
        self.shards.complete_columns(total.packed)
        return total

    def set_degrees(self, total_degree: dict[str, int]) -> None:
        self._degrees = dict(total_degree)
        for side in self._side.values():
            side.set_degrees(self._degrees)

    # -- filters
    def device_filters(self, filters: dict[str, Any] | None) -> list[tuple[int, int]] | None:
        """dict -> [(column, code)]; None when some value was never stored (nothing can match)."""
        out = []
        for key, value in (filters or {}).items():
            if key not in self.keys:
                raise ValueError(f"collection {self.name!r} cannot filter on payload key {key!r} "
                                 f"(filterable: {', '.join(self.keys)})")
            code = self.payloads.cols[key].code_of(value)
            if code is None:
                return None
            out.append((self.keys.index(key), code))
        return out

    def matching_slots(self, filters: dict[str, Any] | None, limit: int | None = None) -> np.ndarray:
        """Alive slots matching every equality, in insertion order -- resolved on the device from the code columns."""
        dfilt = self.device_filters(filters)
        if dfilt is None:
            return np.zeros((0,), np.int64)
        sh, lo = self.shards.match_rows(dfilt, sum(self.shards.rows) if limit is None else limit)
        slots = np.sort(self.slots_of(sh, lo))
        return slots if limit is None else slots[:limit]

    # -- mutation
    def remove_slots(self, slots) -> None:
        slots = np.asarray(slots, dtype=np.int64)
        if slots.size:
            self.shards.tombstone(*self.rows_of(slots))

    def delete(self, filters: dict[str, Any]) -> int:
        """client.py:159-169: every point matching the AND of equalities.  An empty filter matches every point, as
        ``Filter(must=[])`` does."""
        dfilt = self.device_filters(filters)
        if dfilt is None:
            return 0
        if not dfilt:
            slots = self.matching_slots(None)
            self.remove_slots(slots)
            n = int(slots.size)
        else:
            n = self.shards.tombstone_filter(dfilt)
        self.maybe_compact()
        return n

    def upsert(self, ids, vectors, payloads, preprocessed: bool = False, texts=None, embed=None) -> None:
        """``vectors``: list of float lists (what the reference passes), a float32 ndarray [n, dim], or a CUDA tensor [n, dim]
        on this collection's device (no host round trip) -- or None with ``texts`` and ``embed`` (``embed(list[str])`` -> ndarray /
        CUDA tensor [m, dim]): the rows are routed first and every process embeds only the texts of the shards it owns, straight
        into them (embeddings/indexer.py:66-85 with the embedding work sharded like the rows: SURVEY 8(e), embed row)."""
        n = len(ids)
        if n == 0:
            return
        lazy = vectors is None
        if lazy:
            if texts is None or embed is None or len(texts) != n:
                raise ValueError("upsert without vectors needs one text per id and an embed callable")
            vecs, on_dev = None, False
        else:
            on_dev = not isinstance(vectors, (list, tuple, np.ndarray)) and bool(getattr(vectors, "is_cuda", False))
            vecs = vectors if on_dev else np.asarray(vectors, dtype=np.float32)
            if vecs.ndim != 2 or int(vecs.shape[0]) != n:
                raise ValueError(f"upsert needs equally many ids, vectors and payloads (got {n}, {tuple(vecs.shape)}, {len(payloads)})")
            if int(vecs.shape[1]) != self.shards.dim:
                raise ValueError(f"vector dimension {vecs.shape[1]} does not match the collection's {self.shards.dim}")
        if len(payloads) != n:
            raise ValueError(f"upsert needs equally many ids, vectors and payloads (got {n} ids, {len(payloads)} payloads)")
        ids = [str(i) for i in ids]
        last = dict(zip(ids, range(n)))                       # a repeated id inside one call: last one wins
        if len(last) != n:
            keep = sorted(last.values())
            ids, payloads = [ids[i] for i in keep], [payloads[i] for i in keep]
            if lazy:
                texts = [texts[i] for i in keep]
            else:
                vecs = vecs[keep] if not on_dev else vecs[ffi_index_tensor(vecs, keep)]
            n = len(keep)
        n0 = self.payloads.n
        saved_next = self.shards._next_block
        shard = self.shards.route(n)
        if self.partial:                                      # the text of rows other ranks own stays with them
            owned = np.isin(shard, self.shards.owned)
            stored = [p if o else {k: ("" if k in TEXT_KEYS and isinstance(v, str) else v) for k, v in p.items()} for p, o in zip(payloads, owned)]
        else:
            stored = payloads
        self.payloads.extend(stored)
        orphans: dict[int, tuple[int, int]] = {}
        try:
            codes = self.payloads.device_codes(self.keys, n0, n0 + n) if self.keys else None
            if lazy:
                per_shard = {}
                embed_failure = None
                try:
                    for sh in self.shards.owned:
                        sel = np.flatnonzero(shard == sh)
                        if sel.size:
                            v = embed([texts[i] for i in sel])
                            v = v if bool(getattr(v, "is_cuda", False)) else np.asarray(v, dtype=np.float32)
                            if v.ndim != 2 or int(v.shape[0]) != sel.size or int(v.shape[1]) != self.shards.dim:
                                raise ValueError(f"embed returned {tuple(v.shape)} for {sel.size} texts of dimension {self.shards.dim}")
                            per_shard[sh] = v
                except Exception as e:  # noqa: BLE001 -- carried into the append's agreement (below), re-raised from there
                    embed_failure = e
                dev_rows = [v for v in per_shard.values() if not isinstance(v, np.ndarray)]
                if embed_failure is not None:
                    # one rank's encoder failed (a bad text, out of memory): the others are already on their way into the append's
                    # agreement collective -- this rank joins it with its failure, so that everyone rolls back and raises together
                    self.shards.append({}, codes, preprocessed, shard=shard, failure=embed_failure)
                    raise embed_failure          # (not reached: append re-raises it)
                if dev_rows:
                    import torch
                    per_shard = {sh: (v.contiguous() if v.dtype == torch.float32 else v.float().contiguous()) if not isinstance(v, np.ndarray) else v
                                 for sh, v in per_shard.items()}
                    _, local = self.shards.append(per_shard, codes, preprocessed, stream=ffi.current_stream(dev_rows[0].device), shard=shard)
                    torch.cuda.current_stream(dev_rows[0].device).synchronize()
                else:
                    _, local = self.shards.append(per_shard, codes, preprocessed, shard=shard)
            elif on_dev:
                import torch
                # (the caller's tensor comes from the device's default stream -- the encoder's --, this job may run on the store's own)
                torch.cuda.current_stream(vecs.device).wait_stream(torch.cuda.default_stream(vecs.device))
                vecs = vecs.contiguous() if vecs.dtype == torch.float32 else vecs.float().contiguous()
                _, local = self.shards.append(vecs, codes, preprocessed, stream=ffi.current_stream(vecs.device), shard=shard)
                torch.cuda.current_stream(vecs.device).synchronize()      # the caller may free / reuse its tensor right away
            else:
                _, local = self.shards.append(vecs, codes, preprocessed, shard=shard)
        except AppendFailed as e:
            # some shards took their rows before another refused: those rows are dead on the device and have no slot -- the
            # maps step over them (-1), the tables go back
            self.payloads.truncate(n0)
            if self.shards.ns > 1:
                for sh, (_, m) in e.done.items():
                    self.slot_of[sh] = np.concatenate([self.slot_of[sh], np.full((m,), -1, np.int64)])
            else:
                raise RuntimeError("a one-shard collection cannot lose an append half-way") from e
            # the orphans are dead on the device already: reclaim them now, so that no later state (a snapshot, the side columns of
            # the re-rank) ever holds a row without a slot.  Every rank is here (the append agreed on the failure), so the
            # compaction's collectives line up.  Should the compaction itself fail, the maps above keep the collection usable
            # and save() tries again.
            try:
                self.compact()
            except Exception as ce:  # noqa: BLE001
                raise e.cause from ce
            self.shards._next_block = saved_next              # nothing of the call is left: the next rows are routed as if it had not been made
            raise e.cause
        except Exception:
            self.payloads.truncate(n0)                        # nothing was stored: the tables go back to where they were
            self.shards._next_block = saved_next
            raise
        replaced = self.ids.extend(ids)                       # slots these ids occupied before (-1: new)
        if self.shards.ns > 1:
            self.row_shard = np.concatenate([self.row_shard, shard])
            self.row_local = np.concatenate([self.row_local, local])
            for s in range(self.shards.ns):
                m = shard == s
                if m.any():
                    self.slot_of[s] = np.concatenate([self.slot_of[s], n0 + np.flatnonzero(m)])
        stale = replaced[replaced >= 0]
        if stale.size:
            self.remove_slots(stale)                          # (a slot that is already dead is tombstoned again: the device ignores it)
            self.maybe_compact()

    def hit(self, slot: int, score: float) -> dict[str, Any]:
        return {"id": self.ids.get(slot), "score": score, "payload": self.payloads_of([slot])[0]}

    def payloads_of(self, slots) -> list[dict[str, Any]]:
        """The stored payload dictionaries of ``slots`` (every process passes the same list).  With one process per shard only a
        row's OWNER holds its text: each rank serialises the payloads of the slots it owns and ONE byte exchange completes the
        list everywhere (``ShardSet.exchange_bytes``: two tensor all-reduces)."""
        slots = [int(t) for t in slots]
        if not self.partial:
            return [self.payloads.get(t) for t in slots]
        import json
        mine = set(self.shards.owned)
        # only the TEXT fields travel (everything else is replicated and keeps its Python type: tuples, None, numbers)
        local = [self.payloads.get(t) for t in slots]
        parts = [json.dumps({k: p[k] for k in TEXT_KEYS if isinstance(p.get(k), str)}, ensure_ascii=False).encode("utf-8", "surrogatepass")
                 if int(self.row_shard[t]) in mine else None for t, p in zip(slots, local)]
        out = []
        for p, b in zip(local, self.shards.exchange_bytes(parts)):
            p = dict(p)
            p.update(json.loads(b.decode("utf-8", "surrogatepass")))
            out.append(p)
        return out

    def hits(self, slots, scores) -> list[dict[str, Any]]:
        """Hit dictionaries of a flat list of (slot, score) pairs, payloads fetched together."""
        slots = [int(t) for t in slots]
        pays = self.payloads_of(slots)
        return [{"id": self.ids.get(t), "score": float(sc), "payload": p} for t, sc, p in zip(slots, scores, pays)]

    def search(self, queries: np.ndarray, limit: int, dfilt) -> tuple[np.ndarray, np.ndarray]:
        """(scores [nq, limit], slots [nq, limit]); -1 slots are padding."""
        scores, shard, local = self.shards.search(queries, limit, dfilt)
        return scores, self.slots_of(shard, local)

    # -- compaction
    def maybe_compact(self) -> bool:
        rows, alive = self.shards.count()
        dead = rows - alive
        if self.compact_dead_fraction > 0 and dead >= self.compact_min_dead and dead >= self.compact_dead_fraction * rows:
            self.compact()
            return True
        return False

    def compact(self) -> int:
        """Reclaim the rows of deleted points on the device (``crh_index_compact``) and drop their ids / payloads from the host
        tables; returns the number of rows reclaimed.  Slots and rows are renumbered; ids, payloads, filters and search
        results are unchanged."""
        before = self.payloads.n
        maps = self.shards.compact()
        if self.shards.ns == 1:
            o2n = maps[0]
            keep = np.flatnonzero(o2n >= 0)
        else:
            new_local = np.full((before,), -1, np.int64)
            for s, o2n in maps.items():
                m = self.row_shard == s
                new_local[m] = o2n[self.row_local[m]]
            keep = np.flatnonzero(new_local >= 0)
            self.row_shard, self.row_local = self.row_shard[keep], new_local[keep]
            for s in range(self.shards.ns):
                m = np.flatnonzero(self.row_shard == s)
                so = np.empty((m.size,), np.int64)
                so[self.row_local[m]] = m
                self.slot_of[s] = so
        if keep.size != before:
            self.ids.compact(keep)
            self.payloads.compact(keep)
        moved = any(bool((o2n < 0).any()) for o2n in maps.values())      # (rows without a slot -- a half-way append -- count as well)
        if moved:
            for s, side in self._side.items():
                side.select(np.flatnonzero(maps[s][: side.rows] >= 0))
            self.compactions += 1
        return int(before - keep.size)

    # -- persistence
    def save(self, directory: str) -> None:
        """``directory``: the index image of every shard (``ffi.Index.save``: raw, mmap-able, verbatim) + the id and payload
        tables as raw arrays (``tables``) + ``collection.json`` (keys, shard layout, graph degrees).  Written next to the
        target and renamed over it, so a crash mid-save leaves the previous snapshot intact."""
        import json
        import shutil
        if any(bool((so < 0).any()) for so in self.slot_of):      # rows without a slot (a half-way append whose clean-up failed) never reach a snapshot
            self.compact()
        tmp = directory.rstrip("/") + ".tmp"
        primary = self.shards.rank in (None, 0)
        if primary:
            shutil.rmtree(tmp, ignore_errors=True)
            os.makedirs(tmp, exist_ok=True)
        self.shards.barrier()                                     # (every rank sees the directory before writing into it)
        self.shards.save(tmp)
        if self.partial:                                          # a rank's payload table holds the text of its own rows only: one table per rank
            sub = os.path.join(tmp, f"tables{self.shards.rank}")
            os.makedirs(sub, exist_ok=True)
            self.payloads.save(sub)
        if primary:
            self.ids.save(tmp)
            if not self.partial:
                self.payloads.save(tmp)
            self.row_shard.tofile(os.path.join(tmp, "rows.shard.i32"))
            self.row_local.tofile(os.path.join(tmp, "rows.local.i64"))
            with open(os.path.join(tmp, "collection.json"), "w") as f:
                # format 4: says where the payload tables are ("text_tables": "per_rank" = tables{rank}/ hold the text of that
                # rank's rows only, written by one process per shard; "root" = one complete table)
                json.dump({"name": self.name, "keys": list(self.keys), "format": 4, "slots": self.payloads.n, "shards": self.shards.ns,
                           "shard_rows": list(self.shards.rows), "degrees": self._degrees,
                           "text_tables": "per_rank" if self.partial else "root"}, f, default=repr)
        self.shards.barrier()
        if primary:
            shutil.rmtree(directory, ignore_errors=True)
            os.replace(tmp, directory)
        self.shards.barrier()

    def load(self, directory: str) -> None:
        import json
        with open(os.path.join(directory, "collection.json")) as f:
            meta = json.load(f)
        if list(meta["keys"]) != list(self.keys):
            raise ValueError(f"snapshot of {self.name} codes the payload keys {meta['keys']}, this store {list(self.keys)}")
        if int(meta.get("shards", 1)) != self.shards.ns:
            raise ValueError(f"snapshot of {self.name} has {meta.get('shards', 1)} shards, this store {self.shards.ns}")
        self.shards.load(directory)
        n = int(meta["slots"])
        self.ids.load(directory, n)
        per_rank = os.path.join(directory, f"tables{self.shards.rank}") if self.partial else None
        layout = meta.get("text_tables") or ("per_rank" if per_rank and os.path.isdir(per_rank) else "root")
        if layout == "per_rank" and not self.partial:
            raise ValueError(f"snapshot of {self.name} was written by one process per shard (payload text split over tables0..{int(meta.get('shards', 1)) - 1}/): "
                             "load it with shard_backend='dist' on as many ranks")
        # a complete table at the root (format <= 3 of a 'dist' store, or a snapshot of in-process shards) serves a per-rank
        # store as well: it merely holds more text than this rank needs
        self.payloads.load(per_rank if layout == "per_rank" else directory)
        self.row_shard = np.fromfile(os.path.join(directory, "rows.shard.i32"), np.int32)
        self.row_local = np.fromfile(os.path.join(directory, "rows.local.i64"), np.int64)
        if self.payloads.n != n or sum(self.shards.rows) != n or list(self.shards.rows) != [int(v) for v in meta["shard_rows"]]:
            raise ValueError(f"snapshot of {self.name}: {self.shards.rows} rows in the shards, {n} ids, {self.payloads.n} payloads")
        self.slot_of = [np.zeros((0,), np.int64) for _ in range(self.shards.ns)]
        if self.shards.ns > 1:
            for s in range(self.shards.ns):
                m = np.flatnonzero(self.row_shard == s)
                so = np.empty((m.size,), np.int64)
                so[self.row_local[m]] = m
                self.slot_of[s] = so
        self._side = {}
        self._degrees = meta.get("degrees")

    def close(self) -> None:
        self.shards.close()


class RerankHits:
    """Ids and payloads of the slots a ``search_rerank_batch`` output refers to, read while the slots still meant them."""

    def __init__(self, by_slot: dict[int, tuple[Any, dict[str, Any]]], degrees: dict[str, int] | None):
        self._by_slot = by_slot
        self._degrees = degrees

    def hit(self, slot: int, score: float) -> dict[str, Any]:
        pid, payload = self._by_slot[int(slot)]
        return {"id": pid, "score": score, "payload": payload}


def ffi_index_tensor(vecs, keep):
    import torch
    return torch.as_tensor(keep, device=vecs.device, dtype=torch.int64)


class _RawClient:
    """The slice of ``AsyncQdrantClient`` that callers reach through ``QdrantManager.client``
    (health check: client.py:66; admin cleanup: projects/cleanup.py:41-61)."""

    MAX_CODE_COMBINATIONS = 4096

    def __init__(self, store: "HipVectorStore"):
        self._store = store

    async def get_collections(self):
        names = list(self._store._collections)
        return type("CollectionsResponse", (), {"collections": [type("CollectionDescription", (), {"name": n})() for n in names]})()

    async def get_collection(self, collection_name: str) -> CollectionInfo:
        return await self._store.get_collection_info(collection_name)

    @staticmethod
    def _conditions(flt) -> list[tuple[str, str, Any]]:
        """Duck-typed qdrant Filter(must=[FieldCondition(key, match=MatchValue|MatchText)]) -> (key, kind, value)."""
        out = []
        for cond in (getattr(flt, "must", None) or []):
            m = getattr(cond, "match", None)
            if hasattr(m, "text"):
                out.append((cond.key, "text", m.text))
            else:
                out.append((cond.key, "value", getattr(m, "value", None)))
        return out

    def _device_plans(self, col: _Collection, conds) -> list[list[tuple[int, int]]] | None:
        """The conditions as a UNION of device filters ([(column, code)] lists), or None when some condition is on a key the
        device does not code (then the payloads are walked on the host).  MatchValue on a coded key is one code; MatchText on
        a coded key is every code whose VALUE contains the text -- the dictionaries are small (distinct files, not rows)."""
        choices: list[list[tuple[int, int]]] = []
        for key, kind, value in conds:
            if key not in col.keys:
                return None
            c = col.keys.index(key)
            book = col.payloads.cols[key]
            if kind == "value":
                code = book.code_of(value)
                opts = [] if code is None else [(c, code)]
            else:
                opts = [(c, i + 1) for i, v in enumerate(book.values) if isinstance(v, str) and str(value) in v]
            choices.append(opts)
        plans: list[list[tuple[int, int]]] = [[]]
        for opts in choices:
            plans = [p + [o] for p in plans for o in opts]
            if len(plans) > self.MAX_CODE_COMBINATIONS:
                return None
        return plans

    def _host_select(self, col: _Collection, conds) -> np.ndarray:
        """Slots whose payload meets conditions on keys the device does not code: a walk over the alive slots' columns."""
        slots = []
        mine = set(col.shards.owned)
        for t in col.matching_slots(None):
            if col.partial and int(col.row_shard[t]) not in mine:
                continue                                  # (one process per shard: a row's text is with its owner, who answers for it)
            ok = True
            for key, kind, value in conds:
                have = col.payloads.value(int(t), key)
                ok = ok and ((isinstance(have, str) and str(value) in have) if kind == "text" else have == value)
            if ok:
                slots.append(int(t))
        if col.partial:
            parts = col.shards._arrays_everyone({col.shards.rank: np.asarray(slots, np.int64)})
            slots = sorted(int(t) for part in parts.values() for t in part)
        return np.asarray(slots, dtype=np.int64)

    async def count(self, collection_name: str, count_filter=None, exact: bool = True):
        def work():
            col = self._store._col(collection_name)
            conds = self._conditions(count_filter)
            plans = self._device_plans(col, conds)
            if plans is None:
                return int(self._host_select(col, conds).size)
            return sum(col.shards.count_matching(p) for p in plans)      # (plans differ in at least one code: disjoint)
        n = await self._store._run(work)
        return type("CountResult", (), {"count": n})()

    async def delete(self, collection_name: str, points_selector=None):
        flt = getattr(points_selector, "filter", points_selector)

        def work():
            col = self._store._col(collection_name)
            conds = self._conditions(flt)
            plans = self._device_plans(col, conds)
            if plans is None:
                col.remove_slots(self._host_select(col, conds))
            else:
                for p in plans:
                    if p:
                        col.shards.tombstone_filter(p)
                    else:                               # no condition at all: every point
                        col.remove_slots(col.matching_slots(None))
            col.maybe_compact()
        await self._store._run(work)


class HipVectorStore:
    """``QdrantManager`` replacement.  Constructor keeps the reference's positional arguments
    (client.py:19-30: host, port, grpc_port) and ignores them; GPU options are keyword-only."""

    def __init__(self, host: str | None = None, port: int | None = None, grpc_port: int | None = None, *,
                 device: int | None = None, dim: int | None = None, dtype: str | None = None,
                 initial_capacity: int | None = None, search_window_ms: float | None = None, shards: int | None = None,
                 shard_backend: str | None = None, process_group=None, compact_dead_fraction: float | None = None,
                 compact_min_dead: int | None = None, stream: str | None = None, _merge_fn=None):
        s = get_settings()
        self._host, self._port, self._grpc_port = host, port, grpc_port
        self._device = s.hip_device if device is None else device
        # quirk Q3: the reference sizes collections from EMBEDDING_DIMENSIONS (default 1536) although UniXcoder
        # emits 768; here an explicit dim wins, a UniXcoder provider implies 768, else the env value is used.
        if dim is None:
            dim = 768 if s.embedding_provider.startswith("unixcoder") else s.embedding_dimensions
        self._dimensions = dim
        name = (dtype or s.hip_store_dtype).lower()
        if name not in _DTYPES:
            raise VectorStoreError(f"Unknown store dtype {name!r} (use 'f32' or 'bf16')")
        self._dtype = _DTYPES[name]
        self._capacity = initial_capacity or s.hip_initial_capacity
        # Row shards (BASELINE configs[3]: the corpus row-sharded over the GPUs): `shards` handles per collection.  Backend
        # "local": all of them in this process (what a one-GPU box can run); "dist": one process per GPU under
        # torch.distributed -- every rank constructs the same store and makes the same calls, the vectors of a collection are
        # spread over the ranks and every search ends with ONE all-gather + merge (shards.ShardSet).  Default: "dist" when a
        # process group with as many ranks as shards is up, "local" otherwise.
        self._shards = int(shards if shards is not None else s.hip_shards)
        if shard_backend is None:
            shard_backend = s.hip_shard_backend
        if shard_backend == "auto":
            shard_backend = "local"
            if self._shards > 1:
                try:
                    import torch.distributed as dist
                    if dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) == self._shards:
                        shard_backend = "dist"
                except ImportError:
                    pass
        self._shard_backend, self._group, self._merge_fn = shard_backend, process_group, _merge_fn
        self._compact = (s.hip_compact_dead_fraction if compact_dead_fraction is None else compact_dead_fraction,
                         s.hip_compact_min_dead if compact_min_dead is None else compact_min_dead)
        # The stream the store's device work runs on.  "priority" (default): its own HIGH-PRIORITY stream -- the reference's query
        # path runs beside its indexing path in one process (providers/unixcoder_provider.py:260: the encoder on a worker thread),
        # and on the default stream a search queues behind every launch of a forward that is under way (13 ms for a 65 k-token
        # batch; on an equal-priority side stream the device was still seen to leave it there); with priority its kernels get
        # the CUs at the forward's next kernel boundary (~1 ms).  "default": torch's default stream, as in rounds 1-3.  Every
        # job of the store completes on the host before the next starts, so mutations (host-synchronous in the library) and
        # searches stay ordered whichever stream each used.
        self._stream_mode = (stream or s.hip_store_stream).lower()
        self._stream = None                        # torch.cuda.Stream, created by connect()
        self._collections: dict[str, _Collection] = {}
        self._client: _RawClient | None = None
        self._executor: ThreadPoolExecutor | None = None
        self._lock = threading.Lock()
        # Concurrent search() calls (many users, one query each -- how the reference's query path arrives) are coalesced: the
        # calls that queue up while a pass over the corpus is running, for the same collection and filter, share the NEXT pass
        # (up to 64 queries cost what one costs); an idle store serves a lone call at once.  search_window_ms > 0 additionally
        # waits that long before each pass; < 0 (or CODERAG_HIP_SEARCH_WINDOW_MS=-1) turns coalescing off.
        if search_window_ms is None:
            search_window_ms = float(os.environ.get("CODERAG_HIP_SEARCH_WINDOW_MS", "0"))
        # (one process per shard: every search ends in collectives, so all ranks must cut their calls into the SAME passes; how
        # many concurrent calls a pass picks up depends on each rank's own timing -- there, every call is its own pass)
        self._search_coalesce = search_window_ms >= 0 and self._shard_backend != "dist"
        self._search_window_s = max(0.0, search_window_ms) / 1e3
        self._search_pending: dict[tuple, list] = {}
        self._search_drainers: dict[tuple, asyncio.Task] = {}
        self.search_passes = 0                      # corpus passes issued by search() (observability / tests)

    # ------------------------------------------------------------------ plumbing
    async def _run(self, fn, *args):
        """All native work goes through one worker thread: one handle, one thread at a time."""
        if self._executor is None:
            raise VectorStoreError("Client not connected. Call connect() first.")
        loop = asyncio.get_running_loop()

        def guarded():
            with self._lock:
                ffi.use_device(self._device)      # the worker thread starts on device 0 whatever CODERAG_HIP_DEVICE says
                if self._stream is None:
                    return fn(*args)
                import torch
                with torch.cuda.stream(self._stream):      # torch-side work and ffi.current_stream() of the job resolve to the store's stream
                    try:
                        return fn(*args)
                    finally:
                        self._stream.synchronize()           # (a job is complete when it returns: the next one may use another stream)
        return await loop.run_in_executor(self._executor, guarded)

    def _col(self, collection: str) -> _Collection:
        name = collection.value if isinstance(collection, CollectionName) else collection
        if name not in self._collections:
            raise KeyError(f"collection {name!r} does not exist (call create_collections())")
        return self._collections[name]

    # ------------------------------------------------------------------ lifecycle (client.py:32-70)
    async def connect(self) -> None:
        if self._client is not None:
            return
        try:
            ffi.lib()
            if ffi.device_count() <= self._device:
                raise ffi.NativeError(ffi.E_NODEVICE, f"HIP device {self._device} is not visible")
            info = ffi.device_info(self._device)
            self._executor = ThreadPoolExecutor(max_workers=1, thread_name_prefix="hip-store")
            self._client = _RawClient(self)
            if self._stream_mode == "priority":
                try:
                    import torch
                    if torch.cuda.is_available():
                        self._stream = torch.cuda.Stream(device=self._device, priority=-1)
                except ImportError:
                    self._stream = None
            logger.info("Connected to HIP vector store on device %d (%s, %s)", self._device, info["name"], info["arch"])
        except Exception as e:
            self._client = None
            raise VectorStoreError("Failed to connect to Qdrant", cause=e)

    async def close(self) -> None:
        if self._client is None:
            return
        try:
            await self._run(lambda: [c.close() for c in self._collections.values()])
            logger.info("Closed HIP vector store")
        except Exception as e:  # same leniency as client.py:52-55
            logger.warning(f"Error closing HIP vector store: {e}")
        finally:
            self._collections = {}
            self._client = None
            self._stream = None
            if self._executor:
                self._executor.shutdown(wait=True)
                self._executor = None

    @property
    def client(self) -> _RawClient:
        if self._client is None:
            raise VectorStoreError("Client not connected. Call connect() first.")
        return self._client

    async def health_check(self) -> bool:
        try:
            await self.client.get_collections()
            await self._run(ffi.device_info, self._device)
            return True
        except Exception as e:
            logger.warning(f"HIP vector store health check failed: {e}")
            return False

    # ------------------------------------------------------------------ collections (client.py:72-113)
    async def create_collections(self) -> None:
        try:
            _ = self.client

            def work():
                for name in (CollectionName.CODE_CHUNKS.value, CollectionName.SUMMARIES.value):
                    if name not in self._collections:
                        self._collections[name] = _Collection(name, self._dimensions, self._dtype, self._capacity, self._device,
                                                              nshards=self._shards, backend=self._shard_backend, group=self._group,
                                                              merge_fn=self._merge_fn, compact_dead_fraction=self._compact[0],
                                                              compact_min_dead=self._compact[1])
                        self._collections[name].shards.stream = int(self._stream.cuda_stream) if self._stream is not None else 0
                        logger.info(f"Created collection: {name}")
            await self._run(work)
        except Exception as e:
            raise VectorStoreError("Failed to create collections", cause=e)

    async def clear_collections(self) -> None:
        def drop():
            for name in (CollectionName.CODE_CHUNKS.value, CollectionName.SUMMARIES.value):
                col = self._collections.pop(name, None)
                if col is not None:
                    col.close()
        _ = self.client
        await self._run(drop)
        await self.create_collections()

    async def get_collection_info(self, collection: str) -> CollectionInfo:
        try:
            def work():
                col = self._col(collection)
                rows, alive = col.shards.count()
                return CollectionInfo(name=col.name, points_count=alive, vectors_count=alive, indexed_vectors_count=alive,
                                      config={"size": col.shards.dim, "distance": "Cosine", "rows_appended": rows,
                                              "capacity_rows": col.shards.capacity_rows, "shards": col.shards.ns,
                                              "shard_rows": list(col.shards.rows), "shard_backend": col.shards.backend,
                                              "compactions": col.compactions,
                                              "dtype": "bf16" if col.shards.dtype == ffi.DTYPE_BF16 else "f32"})
            return await self._run(work)
        except Exception as e:
            raise VectorStoreError(f"Failed to get collection info for {collection}", cause=e)

    # ------------------------------------------------------------------ data path
    async def upsert(self, collection: str, ids: list[str], vectors, payloads: list[dict[str, Any]], *, texts=None, embed=None) -> None:
        """client.py:115-130.  Same id again replaces the point (Qdrant upsert semantics).  ``vectors``: the reference's
        list of float lists, or -- without the list round trip -- a float32 ndarray / a CUDA tensor [n, dim].
        ``vectors=None`` with ``texts`` and ``embed`` (a callable ``list[str] -> [m, dim]``, e.g. ``provider.embed_texts_sync``):
        the store routes the rows first and THIS process embeds only the texts of the shards it owns -- with one process per GPU
        (``shards=N``, backend "dist") rank g embeds and stores exactly its share, no vector ever crosses ranks."""
        try:
            await self._run(lambda: self._col(collection).upsert(ids, vectors, payloads, texts=texts, embed=embed))
            logger.debug(f"Upserted {len(ids)} vectors to {collection}")
        except Exception as e:
            raise VectorStoreError(f"Failed to upsert vectors to {collection}", cause=e)

    def _search_sync(self, collection: str, queries: np.ndarray, limit: int, filters: dict[str, Any] | None):
        col = self._col(collection)
        dfilt = col.device_filters(filters)
        nq = queries.shape[0]
        if dfilt is None or limit <= 0:
            return col, np.full((nq, max(limit, 0)), -np.inf, np.float32), np.full((nq, max(limit, 0)), -1, np.int64)
        scores, slots = col.search(queries, limit, dfilt)
        return col, scores, slots

    def _search_hits_sync(self, collection: str, queries: np.ndarray, limits, filters: dict[str, Any] | None) -> list[list[dict[str, Any]]]:
        """One pass + the hit dictionaries of every query, built HERE -- inside the worker job, under the store's lock.  Slots
        are positions in the host tables and a compaction renumbers them (the store compacts by itself after deletes and
        replacing upserts): a slot handed back to the event loop could name another point, or none, by the time its payload is
        read.  ``limits``: one int for all queries, or one per query (coalesced callers keep their own prefix)."""
        per = [int(limits)] * queries.shape[0] if isinstance(limits, (int, np.integer)) else [int(v) for v in limits]
        col, scores, slots = self._search_sync(collection, queries, max(per, default=0), filters)
        picked = [[(int(r), float(s)) for s, r in zip(srow[:max(lim, 0)], rrow[:max(lim, 0)]) if r >= 0] for lim, srow, rrow in zip(per, scores, slots)]
        flat = col.hits([t for one in picked for t, _ in one], [sc for one in picked for _, sc in one])      # (payloads fetched together)
        out, at = [], 0
        for one in picked:
            out.append(flat[at:at + len(one)])
            at += len(one)
        return out

    async def search(self, collection: str, query_vector: list[float] | None, limit: int = 10,
                     filters: dict[str, Any] | None = None) -> list[dict[str, Any]]:
        """client.py:132-157: descending cosine, ``[{"id", "score", "payload"}]``.  ``query_vector=None`` is the
        filter-only fetch the context builder issues (quirk Q7): first ``limit`` matching points, score 0.0."""
        try:
            if query_vector is None:
                def fetch():
                    col = self._col(collection)
                    slots = col.matching_slots(filters, limit=limit)
                    return col.hits(slots, [0.0] * len(slots))
                results = await self._run(fetch)
            elif len(query_vector) != self._col(collection).shards.dim:   # (must not fail the pass it would have joined)
                raise ValueError(f"query dim {len(query_vector)} != index dim {self._col(collection).shards.dim}")
            elif limit > ffi.MAX_K:                                      # (likewise: only THIS caller is refused)
                raise ValueError(f"limit {limit} exceeds the index's maximum k of {ffi.MAX_K}")
            elif self._search_coalesce:
                results = await self._search_coalesced(collection, query_vector, limit, filters)
            else:
                q = np.asarray(query_vector, dtype=np.float32).reshape(1, -1)
                self.search_passes += 1
                results = (await self._run(self._search_hits_sync, collection, q, limit, filters))[0]
            logger.debug(f"Found {len(results)} results in {collection}")
            return results
        except Exception as e:
            raise VectorStoreError(f"Failed to search {collection}", cause=e)

    async def _search_coalesced(self, collection: str, query_vector, limit: int, filters: dict[str, Any] | None):
        """One entry of a coalesced pass: queue the query, let the key's drainer run the batch, return this call's slice.
        Calls are grouped by (collection, filter); the pass asks for the largest limit of the group and each caller keeps
        its own prefix (an exact top-k list is a prefix of every longer one)."""
        loop = asyncio.get_running_loop()
        name = collection.value if isinstance(collection, CollectionName) else collection
        key = (name, tuple(sorted((k, repr(v)) for k, v in (filters or {}).items())))
        vec = np.asarray(query_vector, dtype=np.float32).reshape(-1)
        fut: asyncio.Future = loop.create_future()
        self._search_pending.setdefault(key, []).append((vec, int(limit), fut))
        task = self._search_drainers.get(key)
        if task is None or task.done():
            self._search_drainers[key] = loop.create_task(self._drain_searches(key, name, filters))
        return await fut

    async def _drain_searches(self, key, name: str, filters) -> None:
        while self._search_pending.get(key):
            await asyncio.sleep(self._search_window_s)       # (0: one turn of the loop, so calls issued together travel together)
            batch = self._search_pending.pop(key, [])
            if not batch:
                break
            for start in range(0, len(batch), 256):
                part = batch[start:start + 256]
                try:
                    q = np.stack([b[0] for b in part])
                    self.search_passes += (len(part) + 63) // 64
                    per_query = await self._run(self._search_hits_sync, name, q, [b[1] for b in part], filters)
                    for (_, _, fut), hits in zip(part, per_query):
                        if not fut.done():
                            fut.set_result(hits)
                except Exception as e:  # noqa: BLE001 -- every caller of the pass sees the failure (wrapped by search())
                    for _, _, fut in part:
                        if not fut.done():
                            fut.set_exception(e)

    async def search_batch(self, collection: str, query_vectors, limit: int = 10,
                           filters: dict[str, Any] | None = None) -> list[list[dict[str, Any]]]:
        """Batched form of :meth:`search` (not in the reference, which sends one query per RPC): one corpus scan
        serves up to 64 queries."""
        try:
            q = np.asarray(query_vectors, dtype=np.float32)
            return await self._run(self._search_hits_sync, collection, q, limit, filters)
        except Exception as e:
            raise VectorStoreError(f"Failed to search {collection}", cause=e)

    async def set_graph_degrees(self, collection: str, total_degree: dict[str, int]) -> None:
        """``{graph_node_id or entity_name: total_degree}`` for the device re-rank's centrality signal (the reference asks
        Memgraph per query, query/engine.py:348-377; a store that keeps the degrees beside the vectors answers on the device)."""
        await self._run(lambda: self._col(collection).set_degrees(total_degree))

    async def search_rerank_batch(self, collection: str, query_vectors, plans, reranker, limit: int = 20,
                                  filters: dict[str, Any] | None = None):
        """One corpus scan for all queries, then the hybrid re-rank of every candidate list on the device
        (``ranking.device.DeviceReranker``): returns ``(hits, output, slots, scores)`` -- the :class:`RerankOutput`, the host
        copies of the [nq, limit] candidate slots / scores it indexes, and ``hits``: a :class:`RerankHits` whose
        ``hit(slot, score)`` answers for every slot the output refers to (the survivors of each query; the whole list of a query
        the device declined).  Their ids and payloads are read inside this job, under the store's lock: a compaction that runs
        after it renumbers the slots (see ``_search_hits_sync``).  Payloads are still read only for the survivors."""
        import torch
        try:
            def work():
                col = self._col(collection)
                dfilt = col.device_filters(filters)
                q = np.ascontiguousarray(np.asarray(query_vectors, dtype=np.float32))
                nq = q.shape[0]
                dev = torch.device("cuda", self._device)
                if dfilt is not None and limit > 0 and nq > 0:
                    s, r = col.shards.search_device(torch.from_numpy(q).to(dev), limit, dfilt)       # r: GLOBAL rows
                else:
                    s = torch.full((nq, limit), float("-inf"), dtype=torch.float32, device=dev)
                    r = torch.full((nq, limit), -1, dtype=torch.int64, device=dev)
                out = reranker.rank(s, r, col.gather_side(r), plans)
                rows = r.cpu().numpy()
                slots = col.slots_of(np.where(rows >= 0, rows // SHARD_STRIDE, 0), np.where(rows >= 0, rows % SHARD_STRIDE, -1))
                wanted = set()
                for qi in range(nq):
                    c = int(out.count[qi])
                    pos = out.index[qi, :c] if c >= 0 else np.flatnonzero(slots[qi] >= 0)
                    wanted.update(int(t) for t in slots[qi, pos] if t >= 0)
                wanted = sorted(wanted)
                return RerankHits({t: (col.ids.get(t), p) for t, p in zip(wanted, col.payloads_of(wanted))}, col._degrees), out, slots, s.cpu().numpy()
            return await self._run(work)
        except Exception as e:
            raise VectorStoreError(f"Failed to search {collection}", cause=e)

    async def delete(self, collection: str, filters: dict[str, Any]) -> None:
        """client.py:159-169: delete every point matching the AND of equalities."""
        try:
            await self._run(lambda: self._col(collection).delete(filters))
            logger.debug(f"Deleted vectors from {collection} with filters: {filters}")
        except Exception as e:
            raise VectorStoreError(f"Failed to delete from {collection}", cause=e)

    async def file_needs_update(self, collection: str, file_path: str, content_hash: str) -> bool:
        """client.py:178-202: True on a miss, on a different stored hash, and on ANY error."""
        try:
            def work():
                col = self._col(collection)
                slots = col.matching_slots({"file_path": file_path}, limit=1)
                if not len(slots):
                    return True
                return col.payloads.value(int(slots[0]), "content_hash") != content_hash
            return bool(await self._run(work))
        except Exception as e:
            logger.warning(f"Error checking file update status: {e}")
            return True

    async def files_need_update(self, collection: str, files: list[tuple[str, str]]) -> list[bool]:
        """:meth:`file_needs_update` for many ``(file_path, content_hash)`` pairs in ONE job (a batched indexer asks once for a
        whole project instead of once per file); True on a miss, on a different stored hash, and for every file on ANY error."""
        try:
            def work():
                col = self._col(collection)
                out = []
                for file_path, content_hash in files:
                    slots = col.matching_slots({"file_path": file_path}, limit=1)
                    out.append(True if not len(slots) else col.payloads.value(int(slots[0]), "content_hash") != content_hash)
                return out
            return [bool(v) for v in await self._run(work)]
        except Exception as e:
            logger.warning(f"Error checking file update status: {e}")
            return [True] * len(files)

    async def delete_files(self, collection: str, file_paths: list[str]) -> None:
        """``delete(collection, {"file_path": p})`` for many files in ONE job (a path the collection has never stored costs a
        dictionary look-up and no device call); the collection compacts at most once, at the end."""
        try:
            def work():
                col = self._col(collection)
                for p in file_paths:
                    dfilt = col.device_filters({"file_path": p})
                    if dfilt:
                        col.shards.tombstone_filter(dfilt)
                col.maybe_compact()
            await self._run(work)
        except Exception as e:
            raise VectorStoreError(f"Failed to delete from {collection}", cause=e)

    async def compact(self, collection: str | None = None) -> int:
        """Reclaim the rows of deleted points (``crh_index_compact`` + the host tables): what Qdrant's optimizer does in the
        background.  The store also does it by itself once dead rows exceed ``compact_dead_fraction`` of a collection
        (``CODERAG_HIP_COMPACT_DEAD_FRACTION``, 0 = never).  Returns the number of rows reclaimed."""
        try:
            names = [collection.value if isinstance(collection, CollectionName) else collection] if collection else list(self._collections)
            return int(await self._run(lambda: sum(self._col(n).compact() for n in names)))
        except Exception as e:
            raise VectorStoreError("Failed to compact", cause=e)

    # ------------------------------------------------------------------ persistence
    async def save(self, directory: str) -> None:
        """Write every collection to ``directory/<name>/``: the index image verbatim (raw ``tiles.bin`` the scan's layout,
        mmap-able; ``alive.u32``; columnar ``codes.i32``; ``master.f32`` for the f32 store; one sub-directory per shard) + the id
        and payload tables as raw arrays + ``collection.json``.  No pickle, no per-row JSON.  Stands in for the Qdrant volume the reference relies on
        for restarts (docker-compose.yml:42-43): an indexed project can be reloaded without re-embedding."""
        try:
            def work():
                os.makedirs(directory, exist_ok=True)
                for name, col in self._collections.items():
                    col.save(os.path.join(directory, name))
            await self._run(work)
        except Exception as e:
            raise VectorStoreError(f"Failed to save collections to {directory}", cause=e)

    async def load(self, directory: str) -> None:
        """Replace the collections' contents with a snapshot written by :meth:`save`.  The stored image goes back verbatim
        (``crh_index_import``): searches return bit-identical scores and the same ids as before, deleted rows stay deleted."""
        try:
            await self.clear_collections()

            def work():
                for name, col in self._collections.items():
                    sub = os.path.join(directory, name)
                    if os.path.isdir(sub):
                        col.load(sub)
            await self._run(work)
        except Exception as e:
            raise VectorStoreError(f"Failed to load collections from {directory}", cause=e)

    async def __aenter__(self):
        await self.connect()
        return self

    async def __aexit__(self, exc_type, exc_val, exc_tb):
        await self.close()


# the name the reference's call sites import
QdrantManager = HipVectorStore
