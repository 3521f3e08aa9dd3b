"""In-memory stand-in for ``ffi.Index`` built on the CPU oracle -- lets the host-side logic of HipVectorStore
(ids, payload coding, filters, deletes, upsert-replace) run in the CPU test tier.  Test infrastructure only."""
import numpy as np

from oracle import search as orc


class FakeIndex:
    instances = []

    def __init__(self, dim=768, dtype=0, capacity_rows=65536, n_code_cols=0, device=0):
        self.dim, self.dtype, self.n_code_cols, self.device = dim, dtype, n_code_cols, device
        self.capacity_rows = (capacity_rows + 31) // 32 * 32
        self.x = np.zeros((0, dim), np.float32)
        self.codes = np.zeros((0, n_code_cols), np.int32)
        self.alive = np.zeros((0,), np.uint8)
        self.closed = False
        FakeIndex.instances.append(self)

    def reserve(self, capacity_rows):
        self.capacity_rows = max(self.capacity_rows, (capacity_rows + 31) // 32 * 32)

    def append(self, vecs, codes=None, stream=0, preprocessed=False):
        vecs = np.asarray(vecs, np.float32)
        assert len(self.x) + len(vecs) <= self.capacity_rows, "append beyond capacity (store must reserve first)"
        first = len(self.x)
        new = vecs if preprocessed else orc.preprocess(vecs, to_bf16=(self.dtype == 1))
        self.x = np.concatenate([self.x, new]) if len(vecs) else self.x
        if self.n_code_cols:
            self.codes = np.concatenate([self.codes, np.asarray(codes, np.int32).reshape(len(vecs), self.n_code_cols)])
        self.alive = np.concatenate([self.alive, np.ones(len(vecs), np.uint8)])
        return first

    def tombstone(self, rows):
        self.alive[np.asarray(rows, np.int64)] = 0

    def count(self):
        return len(self.x), int(self.alive.sum())

    def search(self, queries, k, filters=None, row_base=0, **kw):
        q = orc.preprocess(np.asarray(queries, np.float32), to_bf16=(self.dtype == 1))
        if len(self.x) == 0:
            return np.full((len(q), k), -np.inf, np.float32), np.full((len(q), k), -1, np.int64)
        s, r = orc.search(self.x, q, k, alive=self.alive, codes=self.codes if self.n_code_cols else None, filters=list(filters or []))
        return s, np.where(r >= 0, r + row_base, r)

    def read_rows(self, first, n):
        return self.x[first:first + n].copy()

    def match_rows(self, filters=None, limit=1):
        ok = self.alive.astype(bool).copy()
        for col, code in (filters or []):
            ok &= self.codes[:, col] == code
        return np.flatnonzero(ok)[:limit].astype(np.int64)

    def _mask(self, filters=None):
        ok = self.alive.astype(bool).copy()
        for col, code in (filters or []):
            ok &= self.codes[:, col] == code
        return ok

    def count_matching(self, filters=None):
        return int(self._mask(filters).sum())

    def tombstone_filter(self, filters):
        assert filters, "a delete needs a filter"
        m = self._mask(filters)
        self.alive[m] = 0
        return int(m.sum())

    def alive_words(self):
        bits = np.zeros(((len(self.alive) + 31) // 32) * 32, np.uint8)
        bits[: len(self.alive)] = self.alive
        return np.packbits(bits, bitorder="little").view(np.uint32).copy()

    def compact(self):
        o2n = np.full(len(self.alive), -1, np.int64)
        keep = np.flatnonzero(self.alive)
        o2n[keep] = np.arange(len(keep))
        self.x, self.codes, self.alive = self.x[keep], self.codes[keep], self.alive[keep]
        return o2n

    def stats(self):
        return {"rows": len(self.x), "fallback_used": 0, "max_query_cands": 0}

    def save(self, directory):
        import os
        os.makedirs(directory, exist_ok=True)
        np.save(os.path.join(directory, "fake_x.npy"), self.x)
        np.save(os.path.join(directory, "fake_alive.npy"), self.alive)
        np.save(os.path.join(directory, "fake_codes.npy"), self.codes)

    def load(self, directory):
        import os
        assert len(self.x) == 0, "load() needs an empty index"
        self.x = np.load(os.path.join(directory, "fake_x.npy"))
        self.alive = np.load(os.path.join(directory, "fake_alive.npy"))
        self.codes = np.load(os.path.join(directory, "fake_codes.npy"))
        self.capacity_rows = max(self.capacity_rows, (len(self.x) + 31) // 32 * 32)

    def close(self):
        self.closed = True
