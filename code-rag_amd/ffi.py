"""ctypes binding of ``libcoderag_hip.so`` (declared in ``include/coderag_hip.h``).

This is the only module that touches the native library.  There is no fallback:
if the shared object is missing or a call fails, a :class:`NativeError` is raised.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("CODERAG_HIP_LIB", PKG_DIR / "lib" / "libcoderag_hip.so"))

OK, E_INVALID, E_HIP, E_CAPACITY, E_NODEVICE, E_INTERNAL = 0, -1, -2, -3, -4, -5
DTYPE_F32, DTYPE_BF16 = 0, 1
NOMINATE_BF16_3, NOMINATE_BF16, NOMINATE_INT8 = 0, 1, 2
MAX_FILTERS, MAX_K = 8, 1024
ABI_VERSION = 4        # CRH_ABI_VERSION of include/coderag_hip.h

# every symbol include/coderag_hip.h declares (tests check the library exports all of them)
EXPORTS = (
    "crh_abi_version", "crh_last_error", "crh_device_count", "crh_device_info",
    "crh_index_create", "crh_index_destroy", "crh_index_append", "crh_index_append_preprocessed", "crh_index_tombstone",
    "crh_index_tombstone_filter", "crh_index_compact", "crh_index_export", "crh_index_import",
    "crh_index_count", "crh_index_clear", "crh_index_reserve", "crh_index_read_rows",
    "crh_search", "crh_search_finish", "crh_search_get_stats", "crh_index_set_tuning",
    "crh_index_set_nomination", "crh_index_get_nomination",
    "crh_merge_topk", "crh_merge_topk_strided", "crh_index_match_rows", "crh_index_set_profiling", "crh_index_get_profile",
    "crh_gemm_bf16_bias", "crh_gemm_bf16_bias_res_ln", "crh_gemm_bf16_res_lnstats", "crh_gemm_bf16_lnin", "crh_layernorm_apply", "crh_gemm_bf16_bias_res32_ln", "crh_attn_fwd_varlen", "crh_embed_ln",
    "crh_masked_mean_pool", "crh_gather_rows_i32", "crh_gather_rows_bytes", "crh_gather_rerank_columns", "crh_rerank_vector",
    "crh_embed_ln_packed", "crh_attn_fwd_packed", "crh_masked_mean_pool_packed", "crh_encoder_finish",
)
# exported by lib/libcoderag_hip_debug.so only (same sources built with -DCRH_ENABLE_DEBUG; tools/ and kernel tests)
DEBUG_EXPORTS = ("crh_debug_gemm_variant", "crh_debug_read_ceiling", "crh_debug_i8_move")
DEBUG_LIB_PATH = Path(os.environ.get("CODERAG_HIP_DEBUG_LIB", PKG_DIR / "lib" / "libcoderag_hip_debug.so"))

RR_NAME_BYTES, RR_MAX_ENTITIES, RR_ENTITY_BYTES = 64, 8, 48


class NativeError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libcoderag_hip error {code}: {message}")
        self.code = code


class Filter(C.Structure):
    _fields_ = [("col", C.c_int32), ("code", C.c_int32)]


class RerankQuery(C.Structure):
    """``crh_rerank_query``: one query's weights and lower-cased entity names."""
    _fields_ = [("vector_weight", C.c_double), ("centrality_weight", C.c_double), ("n_entities", C.c_int32),
                ("entity_len", C.c_int32 * 8), ("entity", (C.c_uint8 * 48) * 8), ("pad_", C.c_int32)]


class RerankColumns(C.Structure):
    """``crh_rerank_columns``: device pointers of the gathered per-candidate side data."""
    _fields_ = [(n, C.c_void_p) for n in ("content_len", "degree", "file_code", "key_code", "node_code", "name_len", "name")]


class SearchStats(C.Structure):
    _fields_ = [("rows", C.c_int64), ("tiles", C.c_int64), ("seed_tiles", C.c_int64),
                ("candidates", C.c_int64), ("max_query_cands", C.c_int64),
                ("fallback_used", C.c_int32), ("batches", C.c_int32)]

    def as_dict(self) -> dict:
        return {name: int(getattr(self, name)) for name, _ in self._fields_}


_lib = None
_debug_lib = None


def _preload_hip_runtime() -> None:
    """libcoderag_hip.so is not linked against a HIP runtime (code-rag_amd/build.sh): it binds to the one already
    in the process.  PyTorch ships its own libamdhip64/libhsa-runtime64; loading a second copy beside it breaks
    device discovery and makes streams / RCCL unshareable, so torch's copy goes in first, globally."""
    candidates = []
    try:
        import torch  # noqa: F401  (plumbing: device memory, streams, torch.distributed)
        candidates.append(Path(torch.__file__).resolve().parent / "lib" / "libamdhip64.so")
    except Exception:  # torch absent: a plain ROCm install is fine for C-ABI-only use
        pass
    candidates += [Path(os.environ.get("ROCM_PATH", "/opt/rocm")) / "lib" / "libamdhip64.so"]
    for cand in candidates:
        if cand.exists():
            C.CDLL(str(cand), mode=C.RTLD_GLOBAL)
            return
    raise NativeError(E_NODEVICE, "no libamdhip64.so found (neither PyTorch-ROCm nor $ROCM_PATH/lib)")


def lib() -> C.CDLL:
    """Load the native library once; raise loudly when it has not been built."""
    global _lib
    if _lib is None:
        _lib = _bind(LIB_PATH, debug=False)
    return _lib


def debug_lib() -> C.CDLL:
    """``libcoderag_hip_debug.so``: the product's sources + the ``crh_debug_*`` entry points.  For tools/ and kernel tests
    only -- nothing in the package calls this.  It is a separate library instance (its own handles and error slot)."""
    global _debug_lib
    if _debug_lib is None:
        _debug_lib = _bind(DEBUG_LIB_PATH, debug=True)
    return _debug_lib


def _bind(path: Path, debug: bool) -> C.CDLL:
    if not path.exists():
        raise NativeError(E_INTERNAL, f"{path} is missing -- build it with "
                          "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc)")
    _preload_hip_runtime()
    L = C.CDLL(str(path))
    vp, i32, i64, f32p = C.c_void_p, C.c_int, C.c_int64, C.c_void_p
    L.crh_abi_version.restype = i32
    L.crh_last_error.restype = C.c_char_p
    L.crh_device_count.argtypes = [C.POINTER(i32)]
    L.crh_device_info.argtypes = [i32, C.c_char_p, i32, C.c_char_p, i32, C.POINTER(i64), C.POINTER(i32)]
    L.crh_index_create.argtypes = [i32, i32, i64, i32, i32, C.POINTER(vp)]
    L.crh_index_destroy.argtypes = [vp]
    L.crh_index_append.argtypes = [vp, i64, f32p, i32, vp, C.POINTER(i64), vp]
    L.crh_index_append_preprocessed.argtypes = [vp, i64, f32p, i32, vp, C.POINTER(i64), vp]
    L.crh_index_tombstone.argtypes = [vp, i64, vp]
    L.crh_index_tombstone_filter.argtypes = [vp, C.POINTER(Filter), i32, C.POINTER(i64)]
    L.crh_index_compact.argtypes = [vp, vp, C.POINTER(i64)]
    L.crh_index_export.argtypes = [vp, i64, i64, vp, vp, vp, vp]
    L.crh_index_import.argtypes = [vp, i64, i64, i64, vp, vp, vp, vp]
    L.crh_index_count.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
    L.crh_index_clear.argtypes = [vp]
    L.crh_index_reserve.argtypes = [vp, i64]
    L.crh_index_read_rows.argtypes = [vp, i64, i64, vp]
    L.crh_search.argtypes = [vp, i32, vp, i32, i32, C.POINTER(Filter), i32, i64, vp, vp, i32, vp]
    L.crh_search_finish.argtypes = [vp, vp]
    L.crh_search_get_stats.argtypes = [vp, C.POINTER(SearchStats)]
    L.crh_index_set_tuning.argtypes = [vp, i32, i32, i32, i32]
    L.crh_index_set_nomination.argtypes = [vp, i32]
    L.crh_index_get_nomination.argtypes = [vp, C.POINTER(i32)]
    L.crh_index_set_profiling.argtypes = [vp, i32]
    L.crh_index_get_profile.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(i64)]
    L.crh_merge_topk.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp]
    L.crh_merge_topk_strided.argtypes = [i32, i32, i32, vp, vp, i64, i64, vp, vp, vp]
    L.crh_index_match_rows.argtypes = [vp, C.POINTER(Filter), i32, i64, vp, C.POINTER(i64)]
    L.crh_gemm_bf16_bias.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.crh_gemm_bf16_bias_res_ln.argtypes = [vp, vp, vp, vp, vp, vp, C.c_float, vp, i32, i32, i32, vp]
    L.crh_gemm_bf16_res_lnstats.argtypes = [vp, vp, vp, vp, vp, vp, C.c_float, vp, vp, vp, i32, i32, i32, vp]
    L.crh_gemm_bf16_lnin.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.crh_layernorm_apply.argtypes = [vp, vp, vp, vp, vp, i32, i32, vp]
    L.crh_gemm_bf16_bias_res32_ln.argtypes = [vp, vp, vp, vp, vp, vp, C.c_float, vp, i32, i32, i32, vp]
    L.crh_attn_fwd_varlen.argtypes = [vp, vp, vp, i32, i32, i32, vp]
    L.crh_embed_ln.argtypes = [vp, vp, vp, vp, vp, vp, C.c_float, i32, vp, vp, i32, i32, i32, vp]
    L.crh_masked_mean_pool.argtypes = [vp, vp, vp, i32, i32, i32, vp]
    L.crh_embed_ln_packed.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_float, i32, vp, vp, i32, i32, i32, i32, vp]
    L.crh_attn_fwd_packed.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.crh_masked_mean_pool_packed.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.crh_encoder_finish.argtypes = [vp]
    L.crh_gather_rows_i32.argtypes = [i64, vp, i64, i64, vp, i32, vp, vp]
    L.crh_gather_rows_bytes.argtypes = [i64, vp, i64, i64, vp, i32, vp, vp]
    L.crh_gather_rerank_columns.argtypes = [i64, vp, i64, i64, vp, vp, vp]
    L.crh_rerank_vector.argtypes = [i32, i32, vp, vp, C.POINTER(RerankColumns), vp, C.c_double, i32, i32, i32, vp, vp, vp, vp, vp, vp]
    if debug or hasattr(L, "crh_debug_gemm_variant"):   # (CODERAG_HIP_LIB may point a tool's whole run at the debug build)
        debug = True
        L.crh_debug_gemm_variant.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp]
        L.crh_debug_read_ceiling.argtypes = [vp, vp]
        L.crh_debug_i8_move.argtypes = [vp]
    for name in EXPORTS + (DEBUG_EXPORTS if debug else ()):
        if name != "crh_last_error":
            getattr(L, name).restype = i32
    if L.crh_abi_version() != ABI_VERSION:
        raise NativeError(E_INTERNAL, f"ABI version mismatch between ffi.py and {path.name}")
    return L


def check(rc: int, L: C.CDLL | None = None) -> None:
    if rc != OK:
        raise NativeError(rc, ((L or lib()).crh_last_error() or b"").decode("utf-8", "replace"))


def use_device(device: int) -> None:
    """Make ``device`` the calling thread's current HIP device.  The stateless entry points (GEMMs, attention, merge, gather,
    re-rank) launch on the CURRENT device, and a worker thread (the provider's and the store's executors) starts on device 0
    whatever the thread that created the tensors had selected."""
    import torch
    if torch.cuda.is_available() and torch.cuda.current_device() != int(device):   # (no device: the native call that follows reports it)
        torch.cuda.set_device(int(device))


def current_stream(device) -> int:
    """torch's current stream on ``device`` as a raw hipStream_t."""
    import torch
    return int(torch.cuda.current_stream(device).cuda_stream)


def device_count() -> int:
    n = C.c_int(0)
    rc = lib().crh_device_count(C.byref(n))
    return int(n.value) if rc == OK else 0


def device_info(device: int = 0) -> dict:
    name, arch = C.create_string_buffer(256), C.create_string_buffer(256)
    hbm, cus = C.c_int64(0), C.c_int(0)
    check(lib().crh_device_info(device, name, 256, arch, 256, C.byref(hbm), C.byref(cus)))
    return {"name": name.value.decode(), "arch": arch.value.decode(), "hbm_bytes": int(hbm.value),
            "cu_count": int(cus.value)}


def _filters(filters) -> tuple:
    filters = list(filters or [])
    if len(filters) > MAX_FILTERS:
        raise NativeError(E_INVALID, f"at most {MAX_FILTERS} filters are supported")
    arr = (Filter * max(1, len(filters)))()
    for i, (col, code) in enumerate(filters):
        arr[i].col, arr[i].code = int(col), int(code)
    return arr, len(filters)


def _ptr(x) -> int:
    """Raw address of a numpy array / torch tensor / int."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    return int(x.data_ptr())  # torch tensor


def _is_dev(x) -> int:
    return int(not isinstance(x, np.ndarray) and getattr(x, "is_cuda", False))


_NP_NAMES = {"float32": np.float32, "int32": np.int32, "int64": np.int64}


def _typed(x, want: str, what: str):
    """The C side sees a bare pointer: refuse anything whose element type or layout it would misread.
    numpy inputs are converted (a copy is fine on the host); device tensors must already be right."""
    if x is None or isinstance(x, int):
        return x
    if isinstance(x, np.ndarray):
        return np.ascontiguousarray(x, dtype=_NP_NAMES[want])
    if str(x.dtype).rsplit(".", 1)[-1] != want:
        raise NativeError(E_INVALID, f"{what} must be {want}, got {x.dtype}")
    if not x.is_contiguous():
        raise NativeError(E_INVALID, f"{what} must be contiguous")
    return x


def _out(x, want: str, what: str, shape):
    if isinstance(x, np.ndarray):
        if x.dtype != _NP_NAMES[want] or not x.flags.c_contiguous:
            raise NativeError(E_INVALID, f"{what} must be a C-contiguous {want} array")
    else:
        _typed(x, want, what)
    if tuple(x.shape) != tuple(shape):
        raise NativeError(E_INVALID, f"{what} has shape {tuple(x.shape)}, expected {tuple(shape)}")
    return x


class Index:
    """Owning wrapper of one ``crh_index`` handle (one collection shard on one GPU)."""

    def __init__(self, dim: int = 768, dtype: int = DTYPE_F32, capacity_rows: int = 65536,
                 n_code_cols: int = 0, device: int = 0):
        self.dim, self.dtype, self.n_code_cols, self.device = dim, dtype, n_code_cols, device
        h = C.c_void_p()
        check(lib().crh_index_create(dim, dtype, capacity_rows, n_code_cols, device, C.byref(h)))
        self._h = h
        self.capacity_rows = (capacity_rows + 31) // 32 * 32

    def close(self) -> None:
        if getattr(self, "_h", None) and _lib is not None:   # (module globals are gone at interpreter shutdown)
            _lib.crh_index_destroy(self._h)
            self._h = None

    __del__ = close

    def _handle(self):
        if not self._h:
            raise NativeError(E_INVALID, "index handle is closed")
        return self._h

    def reserve(self, capacity_rows: int) -> None:
        check(lib().crh_index_reserve(self._handle(), capacity_rows))
        self.capacity_rows = max(self.capacity_rows, (capacity_rows + 31) // 32 * 32)

    def append(self, vecs, codes=None, stream: int = 0, preprocessed: bool = False) -> int:
        """vecs: float32 [n, dim] numpy array (host) or CUDA torch tensor.  Returns the first new row.
        ``preprocessed=True`` stores the rows verbatim (restoring a snapshot made with :meth:`read_rows`)."""
        vecs = _typed(vecs, "float32", "vecs")
        codes = _typed(codes, "int32", "codes")
        n = int(vecs.shape[0])
        if n and int(vecs.shape[1]) != self.dim:
            raise NativeError(E_INVALID, f"vector dim {vecs.shape[1]} != index dim {self.dim}")
        if codes is not None and tuple(codes.shape) != (n, self.n_code_cols):
            raise NativeError(E_INVALID, f"codes shape {tuple(codes.shape)} != ({n}, {self.n_code_cols})")
        if codes is not None and _is_dev(codes) != _is_dev(vecs):
            raise NativeError(E_INVALID, "vecs and codes must live in the same memory space")
        first = C.c_int64(-1)
        fn = lib().crh_index_append_preprocessed if preprocessed else lib().crh_index_append
        check(fn(self._handle(), n, _ptr(vecs), _is_dev(vecs), _ptr(codes), C.byref(first), stream))
        return int(first.value)

    def tombstone(self, rows) -> None:
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        check(lib().crh_index_tombstone(self._handle(), rows.shape[0], rows.ctypes.data))

    def tombstone_filter(self, filters) -> int:
        """Delete every alive row matching all ``(column, code)`` predicates, on the device; returns how many."""
        farr, nf = _filters(filters)
        n = C.c_int64(0)
        check(lib().crh_index_tombstone_filter(self._handle(), farr, nf, C.byref(n)))
        return int(n.value)

    def compact(self) -> np.ndarray:
        """Reclaim the rows of deleted points (``crh_index_compact``): the alive rows move down in their old order.  Returns
        ``old_to_new`` (int64 per old row: its new row number, -1 for a deleted one)."""
        rows, _ = self.count()
        o2n = np.empty((rows,), dtype=np.int64)
        after = C.c_int64(0)
        check(lib().crh_index_compact(self._handle(), o2n.ctypes.data if rows else None, C.byref(after)))
        return o2n

    # ------------------------------------------------------------------ snapshot
    SNAPSHOT_FORMAT = 3                     # 3: inside a 1-KiB piece the 16-byte chunks are ordered [row][half]; 2 (rounds 2-3): [half][row]
    SNAPSHOT_CHUNK_TILES = 1 << 15          # 32768 tiles = 1M rows per transfer (1.5 GiB of tiles at dim 768)

    def save(self, directory: str) -> dict:
        """Write the stored image VERBATIM into ``directory``: ``tiles.bin`` (the tiled bf16 image, raw and mmap-able:
        rows/32 tiles of dim/16 KiB), ``master.f32`` ([rows padded to 32, dim] f32; f32 store only), ``alive.u32`` (one word
        per tile, tombstones included), ``codes.i32`` ([n_code_cols][rows padded to 32] int32, columnar) and ``index.json``.
        No pickle, no compression, no f32 inflation of a bf16 store: 10M x 768 bf16 is 15.36 GB on disk."""
        import json
        rows, alive = self.count()
        ntiles = (rows + 31) // 32
        tile_bytes = self.dim // 16 * 1024
        os.makedirs(directory, exist_ok=True)
        # index.json is what makes a directory a snapshot: it goes away first and comes back last (os.replace), carrying the size
        # of every data file, so a save that died half way leaves no directory load() would accept
        try:
            os.remove(os.path.join(directory, "index.json"))
        except FileNotFoundError:
            pass
        meta = {"format": self.SNAPSHOT_FORMAT, "dim": self.dim, "dtype": "bf16" if self.dtype == DTYPE_BF16 else "f32",
                "rows": rows, "alive": alive, "tiles": ntiles, "n_code_cols": self.n_code_cols, "tile_bytes": tile_bytes,
                "files": {"tiles": "tiles.bin", "alive": "alive.u32", "codes": "codes.i32" if self.n_code_cols else None,
                          "master": "master.f32" if self.dtype == DTYPE_F32 else None}}

        def mm(name, dtype, shape):
            if 0 in shape:
                open(os.path.join(directory, name), "wb").close()
                return None
            return np.memmap(os.path.join(directory, name), dtype=dtype, mode="w+", shape=shape)
        tiles = mm("tiles.bin", np.uint8, (ntiles, tile_bytes))
        al = mm("alive.u32", np.uint32, (ntiles,))
        master = mm("master.f32", np.float32, (ntiles * 32, self.dim)) if self.dtype == DTYPE_F32 else None
        codes = mm("codes.i32", np.int32, (self.n_code_cols, ntiles * 32)) if self.n_code_cols else None
        for t0 in range(0, ntiles, self.SNAPSHOT_CHUNK_TILES):
            nt = min(self.SNAPSHOT_CHUNK_TILES, ntiles - t0)
            cbuf = np.empty((self.n_code_cols, nt * 32), np.int32) if self.n_code_cols else None
            check(lib().crh_index_export(self._handle(), t0, nt, tiles[t0:t0 + nt].ctypes.data,
                                         master[t0 * 32:(t0 + nt) * 32].ctypes.data if master is not None else None,
                                         al[t0:t0 + nt].ctypes.data, cbuf.ctypes.data if cbuf is not None else None))
            if cbuf is not None:
                codes[:, t0 * 32:(t0 + nt) * 32] = cbuf
        for m in (tiles, al, master, codes):
            if m is not None:
                m.flush()
        del tiles, al, master, codes
        meta["sizes"] = {name: os.path.getsize(os.path.join(directory, name)) for name in meta["files"].values() if name}
        with open(os.path.join(directory, "index.json.tmp"), "w") as f:
            json.dump(meta, f)
            f.flush()
            os.fsync(f.fileno())
        os.replace(os.path.join(directory, "index.json.tmp"), os.path.join(directory, "index.json"))
        return meta

    def load(self, directory: str) -> dict:
        """Fill this EMPTY index from a directory written by :meth:`save` (same dim / dtype / code columns): the files are
        memory-mapped and moved to HBM chunk by chunk; searches answer with the same ids and identical score bits."""
        import json
        with open(os.path.join(directory, "index.json")) as f:
            meta = json.load(f)
        want = {"dim": self.dim, "dtype": "bf16" if self.dtype == DTYPE_BF16 else "f32", "n_code_cols": self.n_code_cols}
        for key, val in want.items():
            if meta.get(key) != val:
                raise NativeError(E_INVALID, f"snapshot {directory} has {key}={meta.get(key)!r}, this index {val!r}")
        if meta.get("format") not in (2, self.SNAPSHOT_FORMAT):
            raise NativeError(E_INVALID, f"snapshot {directory} has format={meta.get('format')!r}, this library reads 2 and {self.SNAPSHOT_FORMAT}")
        old_piece_order = meta.get("format") == 2        # [half][row] chunks inside a piece: reordered chunk by chunk below
        if self.count()[0] != 0:
            raise NativeError(E_INVALID, "load() needs an empty index")
        rows, ntiles, tile_bytes = int(meta["rows"]), int(meta["tiles"]), int(meta["tile_bytes"])
        if ntiles != (rows + 31) // 32 or tile_bytes != self.dim // 16 * 1024:
            raise NativeError(E_INVALID, f"snapshot {directory}: inconsistent index.json")
        for name, size in (meta.get("sizes") or {}).items():
            if os.path.getsize(os.path.join(directory, name)) != int(size):
                raise NativeError(E_INVALID, f"snapshot file {os.path.join(directory, name)} is not the size index.json recorded")
        if rows == 0:
            return meta
        self.reserve(rows)

        def mm(name, dtype, shape):
            path = os.path.join(directory, name)
            if os.path.getsize(path) != int(np.prod(shape)) * np.dtype(dtype).itemsize:
                raise NativeError(E_INVALID, f"snapshot file {path} has the wrong size")
            return np.memmap(path, dtype=dtype, mode="r", shape=shape)
        tiles = mm("tiles.bin", np.uint8, (ntiles, tile_bytes))
        al = mm("alive.u32", np.uint32, (ntiles,))
        master = mm("master.f32", np.float32, (ntiles * 32, self.dim)) if self.dtype == DTYPE_F32 else None
        codes = mm("codes.i32", np.int32, (self.n_code_cols, ntiles * 32)) if self.n_code_cols else None
        for t0 in range(0, ntiles, self.SNAPSHOT_CHUNK_TILES):
            nt = min(self.SNAPSHOT_CHUNK_TILES, ntiles - t0)
            cbuf = np.ascontiguousarray(codes[:, t0 * 32:(t0 + nt) * 32]) if codes is not None else None
            abuf = np.ascontiguousarray(al[t0:t0 + nt])
            tbuf = tiles[t0:t0 + nt]
            if old_piece_order:
                tbuf = np.ascontiguousarray(tbuf.reshape(nt, self.dim // 16, 2, 32, 16).transpose(0, 1, 3, 2, 4))
            check(lib().crh_index_import(self._handle(), t0, nt, min(rows, (t0 + nt) * 32), tbuf.ctypes.data,
                                         master[t0 * 32:(t0 + nt) * 32].ctypes.data if master is not None else None,
                                         abuf.ctypes.data, cbuf.ctypes.data if cbuf is not None else None))
        got_rows, got_alive = self.count()
        if got_rows != rows or got_alive != int(meta["alive"]):
            raise NativeError(E_INTERNAL, f"snapshot {directory}: restored {got_rows} rows / {got_alive} alive, index.json says {rows} / {meta['alive']}")
        return meta

    def alive_words(self) -> np.ndarray:
        """One validity word per 32-row tile (bit b of word t = row 32t+b is alive)."""
        ntiles = (self.count()[0] + 31) // 32
        out = np.zeros((ntiles,), np.uint32)
        if ntiles:
            check(lib().crh_index_export(self._handle(), 0, ntiles, None, None, out.ctypes.data, None))
        return out

    def count(self) -> tuple[int, int]:
        r, a = C.c_int64(0), C.c_int64(0)
        check(lib().crh_index_count(self._handle(), C.byref(r), C.byref(a)))
        return int(r.value), int(a.value)

    def clear(self) -> None:
        check(lib().crh_index_clear(self._handle()))

    def read_rows(self, first: int, n: int) -> np.ndarray:
        out = np.empty((n, self.dim), dtype=np.float32)
        check(lib().crh_index_read_rows(self._handle(), first, n, out.ctypes.data))
        return out

    def set_tuning(self, seed_tiles: int = 0, wave_cand_cap: int = 0, query_cand_cap: int = 0,
                   force_fallback: int = -1) -> None:
        check(lib().crh_index_set_tuning(self._handle(), seed_tiles, wave_cand_cap, query_cand_cap, force_fallback))

    def set_nomination(self, mode: int) -> None:
        """The most advanced way the index may nominate a <= 64-query batch: NOMINATE_BF16_3 / NOMINATE_BF16 / NOMINATE_INT8."""
        check(lib().crh_index_set_nomination(self._handle(), mode))

    def nomination(self) -> int:
        """The mode the next <= 64-query batch would use (the index may have fallen back by itself)."""
        out = C.c_int32(0)
        check(lib().crh_index_get_nomination(self._handle(), C.byref(out)))
        return int(out.value)

    def search(self, queries, k: int, filters=None, row_base: int = 0, out_scores=None, out_rows=None,
               stream: int = 0):
        """queries: [nq, dim] float32 numpy (host) or CUDA tensor.  With ``out_*`` CUDA tensors the call is
        asynchronous (finish with :meth:`search_finish`); otherwise numpy results are returned."""
        queries = _typed(queries, "float32", "queries")
        if queries.ndim == 1:
            queries = queries[None, :]
        nq = int(queries.shape[0])
        if nq and int(queries.shape[1]) != self.dim:
            raise NativeError(E_INVALID, f"query dim {queries.shape[1]} != index dim {self.dim}")
        farr, nf = _filters(filters)
        if not 0 < k <= MAX_K:
            raise NativeError(E_CAPACITY, f"k={k} outside 1..{MAX_K}")
        if out_scores is None:
            out_scores = np.empty((nq, k), dtype=np.float32)
            out_rows = np.empty((nq, k), dtype=np.int64)
        else:
            _out(out_scores, "float32", "out_scores", (nq, k))
            _out(out_rows, "int64", "out_rows", (nq, k))
            if _is_dev(out_scores) != _is_dev(out_rows):
                raise NativeError(E_INVALID, "out_scores and out_rows must live in the same memory space")
        check(lib().crh_search(self._handle(), nq, _ptr(queries), _is_dev(queries), k, farr, nf, row_base,
                               _ptr(out_scores), _ptr(out_rows), _is_dev(out_scores), stream))
        return out_scores, out_rows

    def search_finish(self, stream: int = 0) -> None:
        check(lib().crh_search_finish(self._handle(), stream))

    def stats(self) -> dict:
        s = SearchStats()
        check(lib().crh_search_get_stats(self._handle(), C.byref(s)))
        return s.as_dict()

    def set_profiling(self, enable) -> None:
        """True / 1: HIP events around the dominant kernel of every batch; 2: around all three launches of the int8 scan."""
        check(lib().crh_index_set_profiling(self._handle(), int(enable)))

    def profile(self) -> tuple[float, int]:
        """(total ms, launches) of the scan kernel since profiling was enabled (HIP events on its stream)."""
        ms, n = C.c_double(0.0), C.c_int64(0)
        check(lib().crh_index_get_profile(self._handle(), C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def count_matching(self, filters=None) -> int:
        """Number of alive rows matching the filters (resolved on the device)."""
        farr, nf = _filters(filters)
        n = C.c_int64(0)
        check(lib().crh_index_match_rows(self._handle(), farr, nf, 1 << 62, None, C.byref(n)))
        return int(n.value)

    def match_rows(self, filters=None, limit: int = 1) -> np.ndarray:
        farr, nf = _filters(filters)
        out = np.empty((max(limit, 1),), dtype=np.int64)
        n = C.c_int64(0)
        check(lib().crh_index_match_rows(self._handle(), farr, nf, limit, out.ctypes.data, C.byref(n)))
        return out[: int(n.value)].copy()


def _list_stride(x, want: str, what: str, nl: int, nq: int, k: int) -> int:
    """[nl, nq, k] device tensor whose lists are contiguous but may lie apart (views into one gathered buffer)."""
    if str(x.dtype).rsplit(".", 1)[-1] != want:
        raise NativeError(E_INVALID, f"{what} must be {want}, got {x.dtype}")
    if tuple(x.shape) != (nl, nq, k) or (nq * k > 1 and (x.stride(2) != 1 or (nq > 1 and x.stride(1) != k))):
        raise NativeError(E_INVALID, f"{what} must be [nlists, nq, k] with contiguous lists, got shape {tuple(x.shape)} strides {tuple(x.stride())}")
    return int(x.stride(0)) if nl > 1 else nq * k


def merge_topk(scores, rows, out_scores, out_rows, stream: int = 0) -> None:
    """scores/rows: CUDA tensors [nlists, nq, k] (f32 / i64), each list contiguous; out_*: [nq, k]."""
    nl, nq, k = (int(v) for v in scores.shape)
    use_device(scores.device.index)
    ss = _list_stride(scores, "float32", "scores", nl, nq, k)
    rs = _list_stride(rows, "int64", "rows", nl, nq, k)
    _out(out_scores, "float32", "out_scores", (nq, k))
    _out(out_rows, "int64", "out_rows", (nq, k))
    check(lib().crh_merge_topk_strided(nl, nq, k, _ptr(scores), _ptr(rows), ss, rs, _ptr(out_scores), _ptr(out_rows), stream))


def topk_exchange_buffers(torch, world: int, nq: int, k: int, device):
    """Buffers for the cross-shard exchange with ONE collective: every rank's record is [scores f32 nq*k | rows i64 nq*k]
    (the score block padded to 8 bytes), so a single all-gather of ``local`` into ``gathered`` moves both, and the views
    ``all_scores`` / ``all_rows`` ([world, nq, k], lists contiguous, ranks one record apart) feed merge_topk as they are.
    Returns (local, loc_scores, loc_rows, gathered, all_scores, all_rows)."""
    nb_s = (nq * k * 4 + 7) // 8 * 8
    nb = nb_s + nq * k * 8
    local = torch.empty((nb,), dtype=torch.uint8, device=device)
    gathered = torch.empty((world, nb), dtype=torch.uint8, device=device)
    loc_s = local[: nq * k * 4].view(torch.float32).view(nq, k)
    loc_r = local[nb_s:].view(torch.int64).view(nq, k)
    all_s = gathered[:, : nq * k * 4].view(torch.float32).unflatten(1, (nq, k))
    all_r = gathered[:, nb_s:].view(torch.int64).unflatten(1, (nq, k))
    return local, loc_s, loc_r, gathered, all_s, all_r
