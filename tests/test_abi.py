"""The C-ABI library loads on a CPU-only box and exports every symbol include/coderag_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "coderag_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(crh_[a-z0-9_]+)\s*\(", text)))


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
    assert lib.crh_abi_version() == 1


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
