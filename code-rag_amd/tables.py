"""Host-side tables of a collection: point ids and payloads, columnar, sized for the 10^7..10^8 rows the device side holds.

What the reference keeps inside the Qdrant server (point id -> payload JSON, ``embeddings/client.py:115-130``; read back as
``ScoredPoint.id`` / ``.payload`` in ``client.py:150-157`` and by the update check ``client.py:178-202``) lives here, by SLOT
(a point's position in insertion order):

* ``IdTable``  -- 16 raw bytes per id: a canonical UUID string (what ``VectorIndexer`` generates, ``embeddings/indexer.py:77``:
  ``str(uuid4())``) is stored as its 16 bytes, any other string as its md5 with the text kept aside; id -> slot goes through a
  sorted array of 64-bit keys (binary search) plus a small dictionary of the newest ids, not through one Python ``str`` and
  one ``dict`` entry per row.
* ``PayloadTable`` -- one column per payload key of ``CodeChunk.to_payload`` (``embeddings/chunker.py:12-37``): dictionary codes
  for the string keys (the same dictionaries code the device's filter columns), int64 for the line numbers, one
  offsets + utf-8 blob pair for ``content`` / ``summary``; anything that does not fit a column (an unknown key, an unexpected
  type) is kept verbatim in a sparse ``extras`` map, so ``get(slot)`` always returns a dict equal to the one that was stored.

Snapshots are these arrays written raw (``save`` / ``load``): no per-row JSON, no pickle.
"""

from __future__ import annotations

import hashlib
import json
import os
from operator import itemgetter
from typing import Any, Sequence

import numpy as np

_HEX = np.full(256, -1, np.int16)
for _i, _c in enumerate(b"0123456789abcdef"):
    _HEX[_c] = _i
_DASH_AT = (8, 13, 18, 23)
_HEX_AT = np.asarray([i for i in range(36) if i not in _DASH_AT])


def _grow(arr: np.ndarray, need: int) -> np.ndarray:
    """Capacity-doubling growth of the leading axis (contents kept)."""
    if need <= arr.shape[0]:
        return arr
    cap = max(need, 2 * arr.shape[0], 1024)
    new = np.zeros((cap,) + arr.shape[1:], arr.dtype)
    new[: arr.shape[0]] = arr
    return new


def _hashable(v: Any) -> Any:
    try:
        hash(v)
        return v
    except TypeError:
        return repr(v)


def _book_key(v: Any) -> Any:
    """Dictionary key of a payload value: strings as they are; anything else together with its type, so that 1, 1.0 and True
    (equal and equally hashed in Python) keep their own codes and come back as what was stored."""
    return v if type(v) is str else (type(v).__name__, _hashable(v))


def fingerprints(ids: Sequence[str]) -> tuple[np.ndarray, np.ndarray]:
    """(uint8 [n, 16], bool [n]): the 16 bytes of every id and whether they ARE the id (a canonical lower-case UUID string)
    or its md5.  A batch of canonical UUIDs is decoded without a Python loop."""
    n = len(ids)
    fp = np.zeros((n, 16), np.uint8)
    is_uuid = np.zeros((n,), bool)
    if n == 0:
        return fp, is_uuid
    raw = None
    try:
        if set(map(len, ids)) == {36}:
            raw = np.frombuffer("".join(ids).encode("ascii"), np.uint8).reshape(n, 36)
    except UnicodeEncodeError:
        raw = None
    if raw is not None:
        nib = _HEX[raw[:, _HEX_AT]]                                 # [n, 32] nibbles, -1 where not [0-9a-f]
        ok = (nib >= 0).all(axis=1) & (raw[:, _DASH_AT] == ord("-")).all(axis=1)
        val = (nib[:, 0::2].astype(np.int16) << 4) | nib[:, 1::2]
        fp[ok] = val[ok].astype(np.uint8)
        is_uuid[:] = ok
        rest = np.flatnonzero(~ok)
    else:
        rest = range(n)
    for i in rest:
        s = ids[i]
        if len(s) == 36 and all(s[p] == "-" for p in _DASH_AT):
            try:
                h = s.replace("-", "")
                if len(h) == 32 and h == h.lower() and all(c in "0123456789abcdef" for c in h):
                    fp[i] = np.frombuffer(bytes.fromhex(h), np.uint8)
                    is_uuid[i] = True
                    continue
            except ValueError:
                pass
        fp[i] = np.frombuffer(hashlib.md5(s.encode("utf-8", "surrogatepass")).digest(), np.uint8)
    return fp, is_uuid


def _uuid_text(b: bytes) -> str:
    h = b.hex()
    return f"{h[:8]}-{h[8:12]}-{h[12:16]}-{h[16:20]}-{h[20:]}"


class IdTable:
    """Point ids by slot, and id -> newest slot."""

    MERGE_MIN = 1 << 16

    def __init__(self) -> None:
        self.n = 0
        self._fp = np.zeros((0, 16), np.uint8)
        self._uuid = np.zeros((0,), bool)
        self._text: dict[int, str] = {}                   # slot -> id text, for the ids that are not canonical UUIDs
        self._keys = np.zeros((0,), np.uint64)            # sorted 64-bit keys (first 8 bytes of the fingerprint) of slots < _sorted_n
        self._slots = np.zeros((0,), np.int64)
        self._sorted_n = 0
        self._recent: dict[bytes, int] = {}               # fingerprint -> newest slot, for slots >= _sorted_n

    # -- write
    def extend(self, ids: Sequence[str]) -> np.ndarray:
        """Append ids (already de-duplicated by the caller); returns the slot each of them REPLACES (-1: a new id)."""
        fp, is_uuid = fingerprints(ids)
        old = self._lookup_fp(fp, is_uuid, ids)
        n0, n = self.n, len(ids)
        self._fp = _grow(self._fp, n0 + n)
        self._uuid = _grow(self._uuid, n0 + n)
        self._fp[n0:n0 + n] = fp
        self._uuid[n0:n0 + n] = is_uuid
        keys = fp.tobytes()
        recent = self._recent
        for i in range(n):
            recent[keys[16 * i:16 * i + 16]] = n0 + i
        for i in np.flatnonzero(~is_uuid):
            self._text[n0 + int(i)] = ids[int(i)]
        self.n = n0 + n
        if len(recent) > max(self.MERGE_MIN, self._sorted_n // 4):
            self._merge()
        return old

    def _merge(self) -> None:
        """Fold the recent ids into the sorted arrays (one argsort over all slots)."""
        keys = np.ascontiguousarray(self._fp[: self.n, :8]).view(">u8").reshape(-1).astype(np.uint64)
        order = np.argsort(keys, kind="stable")               # stable: equal keys stay in slot order (the newest last)
        self._keys, self._slots = keys[order], order.astype(np.int64)
        self._sorted_n = self.n
        self._recent = {}

    # -- read
    def get(self, slot: int) -> str:
        slot = int(slot)
        return _uuid_text(self._fp[slot].tobytes()) if self._uuid[slot] else self._text[slot]

    def _lookup_fp(self, fp: np.ndarray, is_uuid: np.ndarray, ids: Sequence[str]) -> np.ndarray:
        n = fp.shape[0]
        out = np.full((n,), -1, np.int64)
        if self.n == 0 or n == 0:
            return out
        raw = fp.tobytes()
        recent = self._recent
        if recent:
            for i in range(n):
                s = recent.get(raw[16 * i:16 * i + 16])
                if s is not None:
                    out[i] = s
        if self._sorted_n:
            todo = np.flatnonzero(out < 0)
            if todo.size:
                keys = np.ascontiguousarray(fp[todo, :8]).view(">u8").reshape(-1).astype(np.uint64)
                hi = np.searchsorted(self._keys, keys, side="right") - 1          # the NEWEST slot among equal keys comes last
                cand = np.flatnonzero((hi >= 0) & (self._keys[np.maximum(hi, 0)] == keys))
                for j in cand:                                                    # (walk back over a 64-bit key collision: practically never)
                    p, i = int(hi[j]), int(todo[j])
                    while p >= 0 and self._keys[p] == keys[j]:
                        s = int(self._slots[p])
                        if bytes(self._fp[s]) == raw[16 * i:16 * i + 16]:
                            out[i] = s
                            break
                        p -= 1
        # a fingerprint match of two DIFFERENT kinds (a UUID's bytes against an md5) or of two texts with one md5 is not a match
        for i in np.flatnonzero(out >= 0):
            s = int(out[i])
            if bool(self._uuid[s]) != bool(is_uuid[i]) or (not is_uuid[i] and self._text.get(s) != ids[int(i)]):
                out[i] = -1
        return out

    def lookup(self, ids: Sequence[str]) -> np.ndarray:
        """Newest slot of every id, -1 for an unknown one."""
        fp, is_uuid = fingerprints(ids)
        return self._lookup_fp(fp, is_uuid, ids)

    # -- maintenance
    def compact(self, keep: np.ndarray) -> None:
        """Keep the slots ``keep`` (ascending indices), renumbered 0..len(keep)-1."""
        keep = np.asarray(keep, np.int64)
        new_of = np.full((self.n,), -1, np.int64)
        new_of[keep] = np.arange(keep.size)
        self._fp = np.ascontiguousarray(self._fp[keep])
        self._uuid = np.ascontiguousarray(self._uuid[keep])
        self._text = {int(new_of[s]): t for s, t in self._text.items() if new_of[s] >= 0}
        self.n = int(keep.size)
        self._keys, self._slots, self._sorted_n, self._recent = np.zeros((0,), np.uint64), np.zeros((0,), np.int64), 0, {}
        if self.n:
            self._merge()

    def clear(self) -> None:
        self.__init__()

    def save(self, directory: str) -> None:
        self._fp[: self.n].tofile(os.path.join(directory, "ids.fp.u8"))
        self._uuid[: self.n].astype(np.uint8).tofile(os.path.join(directory, "ids.uuid.u8"))
        with open(os.path.join(directory, "ids.text.json"), "w") as f:
            json.dump({str(k): v for k, v in self._text.items()}, f)

    def load(self, directory: str, n: int) -> None:
        self.__init__()
        fp = np.fromfile(os.path.join(directory, "ids.fp.u8"), np.uint8)
        uu = np.fromfile(os.path.join(directory, "ids.uuid.u8"), np.uint8)
        if fp.size != 16 * n or uu.size != n:
            raise ValueError(f"id table in {directory}: {fp.size // 16} fingerprints / {uu.size} flags for {n} rows")
        self._fp, self._uuid, self.n = fp.reshape(n, 16), uu.astype(bool), n
        with open(os.path.join(directory, "ids.text.json")) as f:
            self._text = {int(k): v for k, v in json.load(f).items()}
        if n:
            self._merge()

    def nbytes(self) -> int:
        return int(self._fp.nbytes + self._uuid.nbytes + self._keys.nbytes + self._slots.nbytes)


# ------------------------------------------------------------------------------------------------ payload columns
ABSENT, NONE = -1, 0            # dictionary codes of "the key is not in the payload" / "its value is None"; values are >= 1
K_ABSENT, K_VALUE, K_NONE, K_OTHER = 0, 1, 2, 3    # kinds of an int / text cell (K_OTHER: the value lives in `extras`)

# the payload schema of the reference (embeddings/chunker.py:12-37, embeddings/indexer.py:126-134 for summaries)
DICT_KEYS = ("file_path", "entity_type", "entity_name", "language", "content_hash", "project_name", "graph_node_id")
INT_KEYS = ("start_line", "end_line")
TEXT_KEYS = ("content", "summary")


class _Missing:
    def __repr__(self) -> str:
        return "<missing>"


_MISSING = _Missing()


class DictColumn:
    """A low-cardinality key: int32 code per slot, value <-> code dictionary (code = position in ``values`` + 1)."""

    def __init__(self) -> None:
        self.codes = np.zeros((0,), np.int32)
        self.values: list[Any] = []
        self.book: dict[Any, int] = {_MISSING: ABSENT, None: NONE}      # (+ the two sentinels, so that map(book.get) resolves them)

    def code_of(self, value: Any) -> int | None:
        """Code of a stored value (None = never stored): what a filter on this key compares with."""
        if value is None:
            return NONE
        return self.book.get(_book_key(value))

    def encode(self, vals: list, n0: int, extras: dict, key: str) -> None:
        self.codes = _grow(self.codes, n0 + len(vals))
        book, values = self.book, self.values
        try:                                    # values seen before (and the two sentinels) resolve in one C-level pass
            out = list(map(book.get, vals))
        except TypeError:                       # an unhashable value somewhere in the batch
            out = [None]
        if None not in out:
            self.codes[n0:n0 + len(vals)] = out
            return
        out = []
        for v in vals:
            if type(v) is str:
                c = book.get(v)
                if c is None:
                    values.append(v)
                    c = book[v] = len(values)
            elif v is _MISSING:
                c = ABSENT
            elif v is None:
                c = NONE
            else:
                k = _book_key(v)      # (an unhashable value is coded by its repr: what a filter on it compares)
                c = book.get(k)
                if c is None:
                    values.append(v)
                    c = book[k] = len(values)
            out.append(c)
        self.codes[n0:n0 + len(vals)] = out

    def get(self, slot: int):
        c = int(self.codes[slot])
        return _MISSING if c == ABSENT else (None if c == NONE else self.values[c - 1])


class IntColumn:
    def __init__(self) -> None:
        self.vals = np.zeros((0,), np.int64)
        self.kind = np.zeros((0,), np.int8)

    def encode(self, vals: list, n0: int, extras: dict, key: str) -> None:
        self.vals = _grow(self.vals, n0 + len(vals))
        self.kind = _grow(self.kind, n0 + len(vals))
        if set(map(type, vals)) == {int}:               # the common case: every payload carries an int
            try:
                self.vals[n0:n0 + len(vals)] = vals
                self.kind[n0:n0 + len(vals)] = K_VALUE
                return
            except OverflowError:
                pass
        for i, v in enumerate(vals):
            if v is _MISSING:
                continue
            if type(v) is int and -(1 << 63) <= v < (1 << 63):
                self.vals[n0 + i] = v
                self.kind[n0 + i] = K_VALUE
            elif v is None:
                self.kind[n0 + i] = K_NONE
            else:
                self.kind[n0 + i] = K_OTHER
                extras.setdefault(n0 + i, {})[key] = v

    def get(self, slot: int):
        k = int(self.kind[slot])
        return int(self.vals[slot]) if k == K_VALUE else (None if k == K_NONE else _MISSING)


class TextColumn:
    """Long strings: one utf-8 blob and an offsets array."""

    def __init__(self) -> None:
        self.off = np.zeros((1,), np.int64)
        self.kind = np.zeros((0,), np.int8)
        self.blob = bytearray()

    def encode(self, vals: list, n0: int, extras: dict, key: str) -> None:
        n = len(vals)
        self.off = _grow(self.off, n0 + n + 1)
        self.kind = _grow(self.kind, n0 + n)
        pos = int(self.off[n0])
        if set(map(type, vals)) == {str}:               # the common case: every payload carries a string
            try:
                parts = [v.encode("utf-8") for v in vals]
            except UnicodeEncodeError:
                parts = [v.encode("utf-8", "surrogatepass") for v in vals]
            lens = np.fromiter(map(len, parts), np.int64, n)
            np.cumsum(lens, out=self.off[n0 + 1:n0 + n + 1])
            self.off[n0 + 1:n0 + n + 1] += pos
            self.kind[n0:n0 + n] = K_VALUE
            self.blob += b"".join(parts)
            return
        parts = []
        for i, v in enumerate(vals):
            if type(v) is str:
                b = v.encode("utf-8", "surrogatepass")
                parts.append(b)
                pos += len(b)
                self.kind[n0 + i] = K_VALUE
            elif v is None:
                self.kind[n0 + i] = K_NONE
            elif v is not _MISSING:
                self.kind[n0 + i] = K_OTHER
                extras.setdefault(n0 + i, {})[key] = v
            self.off[n0 + i + 1] = pos
        self.blob += b"".join(parts)

    def get(self, slot: int):
        k = int(self.kind[slot])
        if k == K_VALUE:
            return bytes(self.blob[int(self.off[slot]):int(self.off[slot + 1])]).decode("utf-8", "surrogatepass")
        return None if k == K_NONE else _MISSING


class PayloadTable:
    """Payload dictionaries by slot, stored by column.  ``get(slot)`` == the dict that was stored."""

    def __init__(self, dict_keys: Sequence[str] = DICT_KEYS) -> None:
        self.n = 0
        self.dict_keys = tuple(dict.fromkeys(tuple(dict_keys)))
        self.cols: dict[str, Any] = {k: DictColumn() for k in self.dict_keys}
        self.cols.update({k: IntColumn() for k in INT_KEYS if k not in self.cols})
        self.cols.update({k: TextColumn() for k in TEXT_KEYS if k not in self.cols})
        self.extras: dict[int, dict[str, Any]] = {}       # slot -> {key: value} for what no column holds
        self.key_order: list[str] = []                    # keys in order of first appearance (the order get() rebuilds dicts in)

    def extend(self, payloads: Sequence[dict]) -> None:
        n0, n = self.n, len(payloads)
        if n == 0:
            return
        seen = set(self.key_order)
        cols, extras = self.cols, self.extras
        for key, col in cols.items():          # one pass per key: columnar from the start
            try:
                vals = list(map(itemgetter(key), payloads))
            except KeyError:
                vals = [p.get(key, _MISSING) for p in payloads]
            col.encode(vals, n0, extras, key)
        known = cols.keys()
        for i, p in enumerate(payloads):
            pk = p.keys()
            if not known >= pk:
                for k, v in p.items():
                    if k not in cols:
                        extras.setdefault(n0 + i, {})[k] = v
            if not seen >= pk:
                for k in p:
                    if k not in seen:
                        seen.add(k)
                        self.key_order.append(k)
        self.n = n0 + n

    def get(self, slot: int) -> dict[str, Any]:
        slot = int(slot)
        ex = self.extras.get(slot)
        out = {}
        for k in self.key_order:
            col = self.cols.get(k)
            if ex is not None and k in ex:
                out[k] = ex[k]
            elif col is not None:
                v = col.get(slot)
                if v is not _MISSING:
                    out[k] = v
        return out

    def value(self, slot: int, key: str, default: Any = None) -> Any:
        """One field of one payload without building the dict (``payload.get(key, default)``)."""
        slot = int(slot)
        ex = self.extras.get(slot)
        if ex is not None and key in ex:
            return ex[key]
        col = self.cols.get(key)
        if col is None:
            return default
        v = col.get(slot)
        return default if v is _MISSING else v

    def truncate(self, n: int) -> None:
        """Forget the slots >= n (an upsert whose device append failed); dictionary entries they introduced stay, harmlessly."""
        if n >= self.n:
            return
        for col in self.cols.values():
            if isinstance(col, TextColumn):
                del col.blob[int(col.off[n]):]
                col.kind[n:self.n] = K_ABSENT
            elif isinstance(col, IntColumn):
                col.kind[n:self.n] = K_ABSENT
        self.extras = {t: v for t, v in self.extras.items() if t < n}
        self.n = n

    def device_codes(self, keys: Sequence[str], lo: int, hi: int) -> np.ndarray:
        """[hi - lo, len(keys)] int32 codes of the slots [lo, hi) for the device's filter columns (missing and None both 0)."""
        out = np.zeros((hi - lo, len(keys)), np.int32)
        for c, key in enumerate(keys):
            out[:, c] = np.maximum(self.cols[key].codes[lo:hi], 0)
        return out

    def compact(self, keep: np.ndarray) -> None:
        keep = np.asarray(keep, np.int64)
        new_of = np.full((self.n,), -1, np.int64)
        new_of[keep] = np.arange(keep.size)
        for col in self.cols.values():
            if isinstance(col, DictColumn):
                col.codes = np.ascontiguousarray(col.codes[keep])
            elif isinstance(col, IntColumn):
                col.vals, col.kind = np.ascontiguousarray(col.vals[keep]), np.ascontiguousarray(col.kind[keep])
            else:
                lens = (col.off[1:self.n + 1] - col.off[: self.n])[keep]
                off = np.zeros((keep.size + 1,), np.int64)
                np.cumsum(lens, out=off[1:])
                # the kept slots come in runs of consecutive slots (a deleted file is a run of dead ones): one slice per run
                cuts = np.flatnonzero(np.diff(keep) != 1) + 1 if keep.size else np.zeros((0,), np.int64)
                first = np.concatenate([[0], cuts]).astype(np.int64) if keep.size else cuts
                last = np.concatenate([cuts, [keep.size]]).astype(np.int64) - 1 if keep.size else cuts
                mv = memoryview(col.blob)
                a, b = col.off[keep[first]] if keep.size else [], col.off[keep[last] + 1] if keep.size else []
                new = bytearray(b"".join(mv[int(x):int(y)] for x, y in zip(a, b)))
                mv.release()
                col.blob, col.off, col.kind = new, off, np.ascontiguousarray(col.kind[keep])
        self.extras = {int(new_of[s]): v for s, v in self.extras.items() if new_of[s] >= 0}
        self.n = int(keep.size)

    # -- snapshot: raw arrays + the small dictionaries as JSON
    def save(self, directory: str) -> None:
        meta = {"n": self.n, "dict_keys": list(self.dict_keys), "key_order": self.key_order, "columns": {}}
        for key, col in self.cols.items():
            base = os.path.join(directory, "col." + key)
            if isinstance(col, DictColumn):
                col.codes[: self.n].tofile(base + ".i32")
                meta["columns"][key] = {"kind": "dict", "values": col.values}
            elif isinstance(col, IntColumn):
                col.vals[: self.n].tofile(base + ".i64")
                col.kind[: self.n].tofile(base + ".kind.i8")
                meta["columns"][key] = {"kind": "int"}
            else:
                col.off[: self.n + 1].tofile(base + ".off.i64")
                col.kind[: self.n].tofile(base + ".kind.i8")
                with open(base + ".blob", "wb") as f:
                    f.write(col.blob)
                meta["columns"][key] = {"kind": "text"}
        meta["extras"] = {str(k): v for k, v in self.extras.items()}
        with open(os.path.join(directory, "payloads.json"), "w") as f:
            json.dump(meta, f, default=repr)

    def load(self, directory: str) -> None:
        with open(os.path.join(directory, "payloads.json")) as f:
            meta = json.load(f)
        self.__init__(meta["dict_keys"])
        n = self.n = int(meta["n"])
        self.key_order = list(meta["key_order"])
        for key, info in meta["columns"].items():
            base = os.path.join(directory, "col." + key)
            if info["kind"] == "dict":
                col = self.cols[key] = DictColumn()
                col.codes = np.fromfile(base + ".i32", np.int32)
                col.values = list(info["values"])
                col.book.update({_book_key(v): i + 1 for i, v in enumerate(col.values)})
                ok = col.codes.size == n
            elif info["kind"] == "int":
                col = self.cols[key] = IntColumn()
                col.vals, col.kind = np.fromfile(base + ".i64", np.int64), np.fromfile(base + ".kind.i8", np.int8)
                ok = col.vals.size == n and col.kind.size == n
            else:
                col = self.cols[key] = TextColumn()
                col.off, col.kind = np.fromfile(base + ".off.i64", np.int64), np.fromfile(base + ".kind.i8", np.int8)
                with open(base + ".blob", "rb") as f:
                    col.blob = bytearray(f.read())
                ok = col.off.size == n + 1 and col.kind.size == n and (n == 0 or int(col.off[-1]) == len(col.blob))
            if not ok:
                raise ValueError(f"payload column {key!r} in {directory} does not hold {n} rows")
        self.extras = {int(k): v for k, v in meta.get("extras", {}).items()}

    def nbytes(self) -> int:
        total = 0
        for col in self.cols.values():
            if isinstance(col, DictColumn):
                total += col.codes.nbytes
            elif isinstance(col, IntColumn):
                total += col.vals.nbytes + col.kind.nbytes
            else:
                total += col.off.nbytes + col.kind.nbytes + len(col.blob)
        return int(total)
