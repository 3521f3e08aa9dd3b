"""The C-ABI library loads on a CPU-only box and exports every symbol include/coderag_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(debug: bool = False):
    """Entry points the header declares: the product part, or the part inside `#ifdef CRH_ENABLE_DEBUG`."""
    text = open(os.path.join(ROOT, "include", "coderag_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    a, b = text.index("#ifdef CRH_ENABLE_DEBUG"), text.index("#endif", text.index("#ifdef CRH_ENABLE_DEBUG"))
    part = text[a:b] if debug else text[:a] + text[b:]
    return sorted(set(re.findall(r"\b(crh_[a-z0-9_]+)\s*\(", part)))


def test_header_and_binding_agree():
    import coderag_amd  # noqa: F401
    from coderag_amd import ffi
    assert sorted(ffi.EXPORTS) == declared_symbols()


def test_library_exports_every_declared_symbol():
    import coderag_amd  # noqa: F401
    from coderag_amd import ffi
    lib = ffi.lib()
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} is declared in the header but not exported"
    assert lib.crh_abi_version() == ffi.ABI_VERSION == 4


def test_product_library_exports_no_debug_entry_point():
    """crh_debug_* (timing ablations, kernel-selection overrides) live in libcoderag_hip_debug.so only."""
    import subprocess
    import coderag_amd  # noqa: F401
    from coderag_amd import ffi
    assert declared_symbols(debug=True) == sorted(ffi.DEBUG_EXPORTS) and all(n.startswith("crh_debug_") for n in ffi.DEBUG_EXPORTS)
    names = subprocess.check_output(["nm", "-D", "--defined-only", str(ffi.LIB_PATH)]).decode()
    assert "crh_debug_" not in names
    exported = sorted(set(re.findall(r"\b(crh_[a-z0-9_]+)\b", names)))
    assert exported == declared_symbols(), "the product library exports exactly what the header declares"
    dbg = ffi.debug_lib()
    for name in declared_symbols() + declared_symbols(debug=True):
        assert hasattr(dbg, name)
    src = "".join(open(os.path.join(dp, f)).read() for dp, _, fs in os.walk(os.path.join(ROOT, "code-rag_amd")) for f in fs if f.endswith(".py") and f != "ffi.py")
    assert "debug_lib" not in src and "crh_debug" not in src, "the package itself must not reach for the debug build"


def test_calls_fail_loudly_without_a_device():
    import coderag_amd  # noqa: F401
    from coderag_amd import ffi
    if ffi.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(ffi.NativeError) as e:
        ffi.Index(768, ffi.DTYPE_F32, 1024)
    assert e.value.code == ffi.E_NODEVICE
    h = ctypes.c_void_p()
    assert ffi.lib().crh_index_create(100, 0, 10, 0, 0, ctypes.byref(h)) == ffi.E_INVALID   # dim not supported
    assert b"dim" in ffi.lib().crh_last_error()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "code-rag_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f"{f} reaches into oracle/"


def test_binding_refuses_pointers_the_c_side_would_misread():
    """The ABI takes bare pointers, so the binding is where a float64 / strided / mis-shaped device tensor has to stop
    (a float64 query tensor once went through as garbage float32 pairs)."""
    import numpy as np
    import torch
    from coderag_amd import ffi
    good = torch.zeros((4, 8), dtype=torch.float32)
    assert ffi._typed(good, "float32", "queries") is good
    with pytest.raises(ffi.NativeError, match="float32"):
        ffi._typed(good.double(), "float32", "queries")
    with pytest.raises(ffi.NativeError, match="contiguous"):
        ffi._typed(good.t(), "float32", "queries")
    assert ffi._typed(np.zeros((2, 3)), "float32", "vecs").dtype == np.float32     # host arrays are converted
    with pytest.raises(ffi.NativeError, match="shape"):
        ffi._out(torch.zeros((4, 7), dtype=torch.int64), "int64", "out_rows", (4, 8))
    with pytest.raises(ffi.NativeError, match="int64"):
        ffi._out(np.zeros((4, 8), np.int32), "int64", "out_rows", (4, 8))


def test_rerank_structs_match_the_header():
    """ctypes mirrors of crh_rerank_query / crh_rerank_columns have the size the C compiler gives the header's structs."""
    import ctypes as C
    import subprocess
    import tempfile
    from coderag_amd import ffi
    src = '#include <stdio.h>\n#include "coderag_hip.h"\nint main(){printf("%zu %zu %d %d %d", sizeof(crh_rerank_query), ' \
          'sizeof(crh_rerank_columns), CRH_RR_NAME_BYTES, CRH_RR_MAX_ENTITIES, CRH_RR_ENTITY_BYTES);return 0;}'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        got = subprocess.check_output([os.path.join(d, "s")]).decode().split()
    assert [int(v) for v in got] == [C.sizeof(ffi.RerankQuery), C.sizeof(ffi.RerankColumns), ffi.RR_NAME_BYTES,
                                     ffi.RR_MAX_ENTITIES, ffi.RR_ENTITY_BYTES]


def test_rerank_query_packing():
    from types import SimpleNamespace as NS
    import ctypes as C
    from coderag_amd import ffi
    from coderag_amd.ranking import RankingConfig
    from coderag_amd.ranking.device import pack_queries
    plans = [NS(primary_intent="find_similar", entities=[NS(name="Repo"), NS(name="repo"), NS(name="Löwe")]),
             NS(primary_intent="unknown", entities=[NS(name="x" * 60)])]
    raw = pack_queries(plans, RankingConfig())
    arr = (ffi.RerankQuery * 2).from_buffer_copy(raw.tobytes())
    assert (arr[0].vector_weight, arr[0].centrality_weight) == (0.8, 0.2)
    assert arr[0].n_entities == 2                                    # the set of lower-cased names
    names = {bytes(arr[0].entity[i][: arr[0].entity_len[i]]).decode() for i in range(2)}
    assert names == {"repo", "löwe"}
    assert arr[1].n_entities == -1                                   # longer than CRH_RR_ENTITY_BYTES: the host ranks it
