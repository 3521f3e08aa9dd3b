"""Device-side hybrid re-rank of vector hits (BASELINE config 5).

``HybridRanker.rank_results`` (``src/lattice/query/ranking/ranker.py:24-54``) walks one query's hits in Python: about
0.65 ms per 100 hits, 40 ms for a 64-query batch whose corpus scan takes 2.4 ms.  For candidate lists that hold vector hits
only (no graph context -- what ``QueryEngine.search`` has when the graph finds nothing, and what BASELINE config 5
measures), the same arithmetic runs in one kernel launch over the whole batch (``crh_rerank_vector``), reading the payload
facts it needs from per-row SIDE COLUMNS that live on the device beside the vectors:

    content_len  len(payload["content"])            -> code_quality       (scorer.py:105-114)
    degree       total_degree of the row's node      -> centrality         (scorer.py:48-54; -1 = unknown to the graph)
    file_code    dictionary code of file_path        -> per-file cap       (ranker.py:204-229)
    key_code     code of file:entity_name:start_line -> merge of duplicates (models.py:55-56, ranker.py:171-202)
    node_code    code of graph_node_id or entity_name-> centrality lookup key
    name         lower-cased entity name, 64 bytes   -> query_entity_match (scorer.py:92-96)

Results are bit-identical to ``HybridRanker`` (f64, same operand order); ``tests/test_rerank_gpu.py`` checks that.  Only
the <= 50 survivors per query are turned into ``RankedResult`` objects on the host.
"""

from __future__ import annotations

import contextlib
import ctypes as C
import struct
import threading
from dataclasses import dataclass
from typing import Any, Sequence

import numpy as np

from .. import ffi
from ..query_types import ResultSource
from .model import RankedResult, RankingConfig, RankingSignal

SIGNALS = (RankingSignal.VECTOR_SIMILARITY.value, RankingSignal.QUERY_ENTITY_MATCH.value,
           RankingSignal.CENTRALITY.value, RankingSignal.CODE_QUALITY.value)


def node_key(payload: dict[str, Any]) -> str:
    """The centrality lookup key of a vector hit: ``graph_node_id or entity_name`` (scorer.py:48-54 through
    ranker.py:253-264, where ``qualified_name`` is the hit's ``graph_node_id``)."""
    return payload.get("graph_node_id") or payload.get("entity_name", "") or ""


def merge_key(payload: dict[str, Any]) -> str:
    return f"{payload.get('file_path', '')}:{payload.get('entity_name', '')}:{payload.get('start_line')}"


class SideColumns:
    """Per-row side data of one collection shard, resident on the device; rows are appended in store order.  Host and device
    copies grow geometrically and an append uploads ONLY the new rows (a re-rank after every upsert does not re-send the
    collection); replacing a whole column (graph degrees) re-sends that column alone."""

    INT_COLS = ("content_len", "degree", "file_code", "key_code", "node_code", "name_len")

    def __init__(self, device: int = 0, books: dict[str, dict[str, int]] | None = None):
        import torch
        self._torch = torch
        self.device = torch.device("cuda", device)
        self.rows = 0
        self._host: dict[str, np.ndarray] = {c: np.zeros((0,), np.int32) for c in self.INT_COLS}
        self._names = np.zeros((0, ffi.RR_NAME_BYTES), np.uint8)
        # value -> code dictionaries of file path / merge key / centrality key; the shards of one collection SHARE them
        # (candidates of different shards are compared by code)
        self._books: dict[str, dict[str, int]] = books if books is not None else {"file": {}, "key": {}, "node": {}}
        self._node_keys: list[str] = []
        self._dev: dict[str, Any] | None = None      # device tensors with capacity >= rows
        self._dev_rows = 0                           # rows already uploaded
        self._dirty: set[str] = set()                # columns replaced wholesale since the last upload

    def _code(self, book: str, value: str) -> int:
        b = self._books[book]
        return b.setdefault(value, len(b) + 1)

    def _grow_host(self, need: int) -> None:
        cap = len(self._names)
        if need <= cap:
            return
        cap = max(need, 2 * cap, 1024)
        for c in self.INT_COLS:
            new = np.zeros((cap,), np.int32)
            new[: self.rows] = self._host[c][: self.rows]
            self._host[c] = new
        names = np.zeros((cap, ffi.RR_NAME_BYTES), np.uint8)
        names[: self.rows] = self._names[: self.rows]
        self._names = names

    def append(self, payloads: Sequence[dict[str, Any] | None]) -> None:
        """Rows in store order; ``None`` (a tombstoned row) keeps the numbering aligned."""
        n = len(payloads)
        if n == 0:
            return
        self._grow_host(self.rows + n)
        h, r0 = self._host, self.rows
        width = ffi.RR_NAME_BYTES
        for j, p in enumerate(payloads):
            p = p or {}
            r = r0 + j
            content = p.get("content")
            name = (p.get("entity_name", "") or "").lower().encode("utf-8")
            h["content_len"][r] = len(content) if content else 0
            h["degree"][r] = -1
            h["file_code"][r] = self._code("file", p.get("file_path", "") or "")
            h["key_code"][r] = self._code("key", merge_key(p))
            nk = node_key(p)
            h["node_code"][r] = self._code("node", nk)
            h["name_len"][r] = len(name)
            self._node_keys.append(nk)
            if name:
                self._names[r, : min(len(name), width)] = np.frombuffer(name[:width], dtype=np.uint8)
        self.rows = r0 + n

    def select(self, keep: np.ndarray) -> None:
        """Keep the rows ``keep`` (ascending), renumbered 0..len(keep)-1: what a compaction of the index does to its rows."""
        keep = np.asarray(keep, np.int64)
        for c in self.INT_COLS:
            self._host[c] = np.ascontiguousarray(self._host[c][: self.rows][keep])
        self._names = np.ascontiguousarray(self._names[: self.rows][keep])
        if self._node_keys:
            self._node_keys = [self._node_keys[int(i)] for i in keep]
        self.rows = int(keep.size)
        self._dev, self._dev_rows = None, 0          # re-uploaded on the next gather
        self._dirty.clear()

    def set_degrees(self, total_degree: dict[str, int]) -> None:
        """``{centrality key: total_degree}`` as the graph reports it; keys it does not know stay at -1."""
        self._host["degree"][: self.rows] = np.fromiter((int(total_degree.get(k, -1)) for k in self._node_keys), np.int32, self.rows)
        self._dirty.add("degree")

    def set_int_column(self, name: str, values) -> None:
        """Bulk load of one column (synthetic corpora: arrays generated without payload dictionaries)."""
        v = np.asarray(values, dtype=np.int32)
        if v.shape != (self.rows,):
            raise ValueError(f"column {name} needs {self.rows} values")
        self._host[name][: self.rows] = v
        self._dirty.add(name)

    @classmethod
    def from_arrays(cls, device: int, **cols) -> "SideColumns":
        """Columns given as arrays (``name`` as uint8 [rows, 64]); no dictionaries are kept."""
        self = cls(device)
        self.rows = int(len(cols["content_len"]))
        self._host = {c: np.ascontiguousarray(cols[c], dtype=np.int32) for c in cls.INT_COLS}
        self._names = np.ascontiguousarray(cols["name"], dtype=np.uint8).reshape(self.rows, ffi.RR_NAME_BYTES)
        return self

    def _resident(self) -> dict[str, Any]:
        t = self._torch
        if self._dev is None or int(self._dev["name"].shape[0]) < self.rows:         # (re)allocate with headroom, keep what is there
            cap = max(self.rows, 2 * (int(self._dev["name"].shape[0]) if self._dev else 0), 1024)
            new = {c: t.zeros((cap,), dtype=t.int32, device=self.device) for c in self.INT_COLS}
            new["name"] = t.zeros((cap, ffi.RR_NAME_BYTES), dtype=t.uint8, device=self.device)
            if self._dev is not None and self._dev_rows:
                for c in new:
                    new[c][: self._dev_rows].copy_(self._dev[c][: self._dev_rows])
            self._dev = new
        lo, hi = self._dev_rows, self.rows
        if hi > lo:
            for c in self.INT_COLS:
                self._dev[c][lo:hi].copy_(t.from_numpy(self._host[c][lo:hi]))
            self._dev["name"][lo:hi].copy_(t.from_numpy(self._names[lo:hi]))
            self._dev_rows = hi
        for c in self._dirty:
            self._dev[c][: self.rows].copy_(t.from_numpy(self._host[c][: self.rows]))
        self._dirty.clear()
        return self._dev

    def gather(self, rows_dev, row_base: int = 0, stream: int | None = None) -> dict[str, Any]:
        """Side data of a candidate table ``rows_dev`` (int64 CUDA tensor [nq, k]); rows of other shards give zeros
        (degree: 0 as well, so that the shards' gathers add up -- the owner contributes the real value).
        ``stream=None``: torch's current stream on this device (where ``rows_dev`` was produced)."""
        t = self._torch
        ffi.use_device(self.device.index)
        stream = ffi.current_stream(self.device) if stream is None else stream
        dev = self._resident()
        n = int(rows_dev.numel())
        # all columns are views of ONE int32 buffer ([n] per integer column, then 64 name bytes = 16 words per candidate), so
        # that a sharded caller completes the table with one all-reduce of ``out.packed`` (sharded.gather_columns)
        words = ffi.RR_NAME_BYTES // 4
        buf = t.empty((n * (len(self.INT_COLS) + words),), dtype=t.int32, device=self.device)
        out = PackedColumns()
        out.packed = buf
        # one launch for the seven columns (crh_gather_rerank_columns; six launches of crh_gather_rows_* until round 4)
        cc = ffi.RerankColumns(*(dev[c].data_ptr() for c in (*self.INT_COLS, "name")))
        ffi.check(ffi.lib().crh_gather_rerank_columns(n, rows_dev.data_ptr(), row_base, self.rows, C.byref(cc), buf.data_ptr(), stream))
        for i, c in enumerate(self.INT_COLS):
            out[c] = buf[i * n:(i + 1) * n]
        out["name"] = buf[len(self.INT_COLS) * n:].view(t.uint8).view(n, ffi.RR_NAME_BYTES)
        return out


class PackedColumns(dict):
    """Column name -> tensor, all of them views of ``packed`` (one int32 buffer).  At most one shard contributes a non-zero
    value per element, so summing the buffers of all shards word by word completes every column, the name bytes included."""
    packed: Any = None


@dataclass
class RerankOutput:
    """Per query, in final order: ``index[q, :count[q]]`` are positions in the candidate list; ``count[q] == -1`` means the
    device declined the query (more than 8 entities, an entity name longer than 64 bytes) and the host ranker must run."""
    index: np.ndarray     # int32 [nq, max_total]
    score: np.ndarray     # float64 [nq, max_total]
    signals: np.ndarray   # float64 [nq, max_total, 4] in SIGNALS order
    count: np.ndarray     # int32 [nq]
    hybrid: np.ndarray    # bool [nq, max_total]


_Q_SIZE = C.sizeof(ffi.RerankQuery)
_Q_HEAD = struct.Struct(f"<ddi{ffi.RR_MAX_ENTITIES}i")          # weights, n_entities, entity_len[]
_Q_ENT = ffi.RerankQuery.entity.offset
assert _Q_HEAD.size == _Q_ENT and _Q_ENT + ffi.RR_MAX_ENTITIES * ffi.RR_ENTITY_BYTES + 4 == _Q_SIZE


def pack_queries(plans, config: RankingConfig, weights: dict | None = None) -> np.ndarray:
    """``crh_rerank_query`` array (as bytes) from query plans: the intent's weights and the set of lower-cased entity names
    (ranker.py:33-35).  Too many or too long entities are signalled with n_entities = -1.  (struct.pack_into + slice copies
    into one bytearray: field-by-field ctypes or numpy-record assignment cost 7 us per plan -- more than the kernel takes for
    the whole batch.)"""
    buf = bytearray(_Q_SIZE * len(plans))
    weights = {} if weights is None else weights      # intent -> (vector_weight, centrality_weight); a caller may keep it
    pad = [0] * ffi.RR_MAX_ENTITIES
    for i, plan in enumerate(plans):
        w = weights.get(plan.primary_intent)
        if w is None:
            ww = config.weights_for(plan.primary_intent)
            w = weights[plan.primary_intent] = (ww["vector_weight"], ww["centrality_weight"])
        enc = [n.encode("utf-8") for n in sorted({e.name.lower() for e in plan.entities})]
        base = i * _Q_SIZE
        lens = [len(b) for b in enc]
        if len(enc) > ffi.RR_MAX_ENTITIES or (lens and max(lens) > ffi.RR_ENTITY_BYTES):
            _Q_HEAD.pack_into(buf, base, w[0], w[1], -1, *pad)
            continue
        _Q_HEAD.pack_into(buf, base, w[0], w[1], len(enc), *(lens + pad[len(enc):]))
        for j, b in enumerate(enc):
            o = base + _Q_ENT + j * ffi.RR_ENTITY_BYTES
            buf[o:o + len(b)] = b
    return np.frombuffer(buf, dtype=np.uint8)


class DeviceReranker:
    """``HybridRanker`` for batches of vector-only candidate lists, on the device."""

    def __init__(self, config: RankingConfig | None = None, centrality_top: int = 5, device: int = 0):
        import torch
        self._torch = torch
        self.config = config or RankingConfig()
        self.centrality_top = centrality_top   # QueryEngine looks up the first 5 vector hits (engine.py:358-362)
        self.device = torch.device("cuda", device)
        self._lock = threading.Lock()          # the cached device / pinned output buffers below serve one call at a time

    def rank(self, scores_dev, rows_dev, cols: dict[str, Any], plans, stream: int | None = None) -> RerankOutput:
        with self._lock:
            return self._rank(scores_dev, rows_dev, cols, plans, stream)()

    def rank_async(self, scores_dev, rows_dev, cols: dict[str, Any], plans, stream: int | None = None):
        """:meth:`rank` in two halves: everything is enqueued on ``stream`` (the copy back included) and a callable is returned
        that waits for it and builds the :class:`RerankOutput`.  Between the two the caller may enqueue more work on the stream
        -- a serving loop enqueues the next search there, so that the device has something to do while the host collects this
        batch.  ONE call may be outstanding per reranker (its device and pinned buffers are reused), from one thread."""
        return self._rank(scores_dev, rows_dev, cols, plans, stream)

    def _rank(self, scores_dev, rows_dev, cols: dict[str, Any], plans, stream: int | None = None):
        t = self._torch
        ffi.use_device(self.device.index)
        stream = ffi.current_stream(self.device) if stream is None else stream
        nq, k = (int(v) for v in scores_dev.shape)
        if len(plans) != nq:
            raise ValueError(f"{len(plans)} plans for {nq} candidate lists")
        mt = self.config.max_total
        # ONE device buffer and ONE pinned host mirror hold every output (f64 parts first: alignment), cached per shape; the
        # packed queries go up from pinned memory: one H2D, the kernel, one D2H and one wait per call
        # (five allocations + fills and five synchronising .cpu() copies were 0.3 ms of a 0.43 ms call)
        n = nq * mt
        nbytes = 8 * n + 32 * n + 4 * n + 4 * n + 4 * nq
        ws = self.__dict__.setdefault("_ws", {})
        key = (nq, mt)
        if key not in ws:
            ws[key] = (t.empty((nbytes,), dtype=t.uint8, device=self.device), t.empty((nbytes,), dtype=t.uint8, pin_memory=True))
        dbuf, hbuf = ws[key]
        o_sc = dbuf[:8 * n].view(t.float64).view(nq, mt)
        o_sig = dbuf[8 * n:40 * n].view(t.float64).view(nq, mt, 4)
        o_idx = dbuf[40 * n:44 * n].view(t.int32).view(nq, mt)
        o_flg = dbuf[44 * n:48 * n].view(t.int32).view(nq, mt)
        o_cnt = dbuf[48 * n:].view(t.int32)
        packed = pack_queries(plans, self.config, self.__dict__.setdefault("_weights", {}))
        qkey = ("q", packed.nbytes)
        if qkey not in ws:
            ws[qkey] = (t.empty((packed.nbytes,), dtype=t.uint8, device=self.device), t.empty((packed.nbytes,), dtype=t.uint8, pin_memory=True))
        qdev, qhost = ws[qkey]
        # torch's own work for this call (copies, fills) must sit on the stream the kernel is launched on
        cur = t.cuda.current_stream(self.device)
        on = contextlib.nullcontext() if stream == cur.cuda_stream else t.cuda.stream(t.cuda.ExternalStream(stream, device=self.device))
        with on:
            qhost.numpy()[:] = packed       # (every call ends synchronised: the pinned buffers are free again)
            qdev.copy_(qhost, non_blocking=True)     # (the kernel pads the slots behind a query's survivors itself: no fills here)
            cc = ffi.RerankColumns(*(cols[c].data_ptr() for c in ("content_len", "degree", "file_code", "key_code", "node_code", "name_len", "name")))
            ffi._typed(scores_dev, "float32", "scores")
            ffi._typed(rows_dev, "int64", "rows")
            ffi.check(ffi.lib().crh_rerank_vector(nq, k, scores_dev.data_ptr(), rows_dev.data_ptr(), C.byref(cc), qdev.data_ptr(),
                                                  self.config.entity_match_bonus, self.config.max_per_file, mt, self.centrality_top,
                                                  o_idx.data_ptr(), o_sc.data_ptr(), o_sig.data_ptr(), o_cnt.data_ptr(), o_flg.data_ptr(), stream))
            hbuf.copy_(dbuf, non_blocking=True)
            done = t.cuda.Event()
            done.record()

        def collect() -> RerankOutput:
            done.synchronize()
            h = hbuf.numpy()
            return RerankOutput(h[40 * n:44 * n].view(np.int32).reshape(nq, mt).copy(), h[:8 * n].view(np.float64).reshape(nq, mt).copy(),
                                h[8 * n:40 * n].view(np.float64).reshape(nq, mt, 4).copy(), h[48 * n:].view(np.int32).copy(),
                                h[44 * n:48 * n].view(np.int32).reshape(nq, mt).astype(bool))
        return collect

    @staticmethod
    def materialise(out: RerankOutput, q: int, hits: Sequence[dict[str, Any]]) -> list[RankedResult]:
        """The survivors of query ``q`` as ``RankedResult`` objects, built from the flattened hit dicts of its candidate
        list exactly as ``HybridRanker._from_vector_hit`` builds them (fields of an absorbed duplicate are not filled in --
        the device keeps no payload text; a caller that needs them uses the host ranker)."""
        res = []
        for s in range(int(out.count[q])):
            hit = hits[int(out.index[q, s])]
            res.append(RankedResult(
                file_path=hit.get("file_path", ""), entity_name=hit.get("entity_name", ""), entity_type=hit.get("entity_type", ""),
                qualified_name=hit.get("graph_node_id"), content=hit.get("content"), summary=hit.get("summary"),
                start_line=hit.get("start_line"), end_line=hit.get("end_line"), graph_node_id=hit.get("graph_node_id"),
                final_score=float(out.score[q, s]),
                signal_scores={name: float(out.signals[q, s, j]) for j, name in enumerate(SIGNALS)},
                source=ResultSource.HYBRID.value if out.hybrid[q, s] else ResultSource.VECTOR.value))
        return res
