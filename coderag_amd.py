"""Import shim: makes the package directory ``code-rag_amd/`` importable as ``coderag_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "code-rag_amd")
_spec = importlib.util.spec_from_file_location(
    "coderag_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["coderag_amd"] = _mod
_spec.loader.exec_module(_mod)
