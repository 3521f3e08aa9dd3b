"""CPU oracle for the vector-store search path -- TEST INFRASTRUCTURE ONLY.

Python face of ``oracle/search_oracle.c`` (see its header for the reference
call sites it follows: ``src/lattice/embeddings/client.py:93-176`` and Qdrant's
published scalar cosine algorithm).  Importable only from ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.

PARITY UNPINNED at the Qdrant boundary (no golden vector exists in the
reference, the server image is unpinned and absent): the committed fixtures pin
this restatement against an independent fp64 evaluation, not against Qdrant.
"""

from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "_build" / "liborc_search.so"
_lib = None


def build(force: bool = False) -> Path:
    """Compile the C oracle with gcc (seconds).  Building the checker is not using it."""
    src = _HERE / "search_oracle.c"
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-s", "-C", str(_HERE)], check=True)
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(str(_LIB_PATH))
        f32p = ctypes.POINTER(ctypes.c_float)
        i64p = ctypes.POINTER(ctypes.c_int64)
        i32p = ctypes.POINTER(ctypes.c_int32)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        lib.orc_bf16_round.restype = ctypes.c_float
        lib.orc_bf16_round.argtypes = [ctypes.c_float]
        lib.orc_preprocess_rows.restype = None
        lib.orc_preprocess_rows.argtypes = [f32p, f32p, ctypes.c_int64, ctypes.c_int, ctypes.c_int]
        lib.orc_dot.restype = ctypes.c_float
        lib.orc_dot.argtypes = [f32p, f32p, ctypes.c_int]
        lib.orc_search.restype = ctypes.c_int
        lib.orc_search.argtypes = [f32p, ctypes.c_int64, ctypes.c_int, u8p, i32p, ctypes.c_int,
                                   i32p, i32p, ctypes.c_int, f32p, ctypes.c_int, ctypes.c_int,
                                   f32p, i64p]
        lib.orc_merge_topk.restype = ctypes.c_int
        lib.orc_merge_topk.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, f32p, i64p, f32p, i64p]
        _lib = lib
    return _lib


def _p(a, ct):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ct))


def preprocess(vectors: np.ndarray, to_bf16: bool = False) -> np.ndarray:
    """Qdrant ``cosine_preprocess`` per row (+ optional bf16 rounding of the result)."""
    v = np.ascontiguousarray(vectors, dtype=np.float32)
    if v.ndim == 1:
        v = v[None, :]
    out = np.empty_like(v)
    _load().orc_preprocess_rows(_p(v, ctypes.c_float), _p(out, ctypes.c_float), v.shape[0], v.shape[1],
                                int(to_bf16))
    return out


def dot(a: np.ndarray, b: np.ndarray) -> float:
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return float(_load().orc_dot(_p(a, ctypes.c_float), _p(b, ctypes.c_float), a.shape[0]))


def search(corpus_pre: np.ndarray, queries_pre: np.ndarray, k: int, alive: np.ndarray | None = None,
           codes: np.ndarray | None = None, filters: list[tuple[int, int]] | None = None,
           threads: int | None = None):
    """Exact top-k on already preprocessed corpus/queries.  Returns (scores[nq,k] f32, rows[nq,k] i64)."""
    x = np.ascontiguousarray(corpus_pre, dtype=np.float32)
    q = np.ascontiguousarray(queries_pre, dtype=np.float32)
    if q.ndim == 1:
        q = q[None, :]
    n, d = x.shape if x.size else (0, q.shape[1])
    nq = q.shape[0]
    out_s = np.empty((nq, k), dtype=np.float32)
    out_r = np.empty((nq, k), dtype=np.int64)
    al = None if alive is None else np.ascontiguousarray(alive, dtype=np.uint8)
    cd = None if codes is None else np.ascontiguousarray(codes, dtype=np.int32)
    ncols = 0 if cd is None else cd.shape[1]
    filters = filters or []
    fc = np.asarray([f[0] for f in filters], dtype=np.int32)
    fv = np.asarray([f[1] for f in filters], dtype=np.int32)
    if threads is not None:
        os.environ["OMP_NUM_THREADS"] = str(threads)
    rc = _load().orc_search(_p(x, ctypes.c_float), n, d, _p(al, ctypes.c_uint8), _p(cd, ctypes.c_int32), ncols,
                            _p(fc, ctypes.c_int32), _p(fv, ctypes.c_int32), len(filters),
                            _p(q, ctypes.c_float), nq, k, _p(out_s, ctypes.c_float), _p(out_r, ctypes.c_int64))
    if rc != 0:
        raise ValueError("orc_search rejected its arguments")
    return out_s, out_r


def cosine_search(vectors: np.ndarray, queries: np.ndarray, k: int, bf16: bool = False, **kw):
    """Raw vectors in, reference semantics out: preprocess on insert and on query, then top-k."""
    return search(preprocess(vectors, bf16), preprocess(queries, bf16), k, **kw)


def merge_topk(scores: np.ndarray, rows: np.ndarray):
    """scores/rows: [nlists, nq, k] -> merged ([nq,k], [nq,k]) in the oracle's total order."""
    s = np.ascontiguousarray(scores, dtype=np.float32)
    r = np.ascontiguousarray(rows, dtype=np.int64)
    nl, nq, k = s.shape
    out_s = np.empty((nq, k), dtype=np.float32)
    out_r = np.empty((nq, k), dtype=np.int64)
    _load().orc_merge_topk(nl, nq, k, _p(s, ctypes.c_float), _p(r, ctypes.c_int64),
                           _p(out_s, ctypes.c_float), _p(out_r, ctypes.c_int64))
    return out_s, out_r


def search_fp64(corpus_pre: np.ndarray, queries_pre: np.ndarray, k: int, alive: np.ndarray | None = None):
    """Independent numpy fp64 evaluation used to pin the C restatement (different code, different precision)."""
    x = np.asarray(corpus_pre, dtype=np.float64)
    q = np.atleast_2d(np.asarray(queries_pre, dtype=np.float64))
    s = q @ x.T
    if alive is not None:
        s[:, ~np.asarray(alive, dtype=bool)] = -np.inf
    order = np.lexsort((np.broadcast_to(np.arange(x.shape[0]), s.shape), -s), axis=1)[:, :k]
    top = np.take_along_axis(s, order, axis=1)
    rows = np.where(np.isneginf(top), -1, order).astype(np.int64)
    pad = k - rows.shape[1]
    if pad > 0:
        top = np.concatenate([top, np.full((top.shape[0], pad), -np.inf)], axis=1)
        rows = np.concatenate([rows, np.full((rows.shape[0], pad), -1, np.int64)], axis=1)
    return top, rows


def search_blas(corpus_pre: np.ndarray, queries_pre: np.ndarray, k: int):
    """Fast CPU exact scan (BLAS sgemm + argpartition): bench.py's cpu_baseline leg, all host cores.

    Same results as :func:`search` up to fp32 summation order; it stands in for "Qdrant exact scan"
    when timing the CPU side.
    """
    s = np.asarray(queries_pre, dtype=np.float32) @ np.asarray(corpus_pre, dtype=np.float32).T
    k = min(k, s.shape[1])
    part = np.argpartition(-s, k - 1, axis=1)[:, :k]
    ps = np.take_along_axis(s, part, axis=1)
    order = np.lexsort((part, -ps), axis=1)
    return np.take_along_axis(ps, order, axis=1), np.take_along_axis(part, order, axis=1).astype(np.int64)
