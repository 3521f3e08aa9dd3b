"""The QdrantManager-surface scenarios on the real HIP index (f32 store: ids and scores equal the f32 oracle)."""
import asyncio

import pytest

pytestmark = pytest.mark.gpu


def test_store_scenarios_on_hip_index(gpu):
    import coderag_amd  # noqa: F401
    from coderag_amd.store import HipVectorStore
    from tests.store_scenarios import run_store_scenarios
    asyncio.run(run_store_scenarios(HipVectorStore(dim=768, dtype="f32", initial_capacity=64, device=0)))


def test_reference_test_database_scenario_dim_1536(gpu, monkeypatch):
    """/root/reference/tests/test_database.py:64-124 on the HIP store at the reference's default EMBEDDING_DIMENSIONS = 1536
    (the store takes its dimension from the settings exactly as QdrantManager does: client.py:29)."""
    import coderag_amd  # noqa: F401
    from coderag_amd.store import CollectionName, QdrantManager
    from tests.store_scenarios import run_reference_database_scenario
    monkeypatch.delenv("EMBEDDING_DIMENSIONS", raising=False)
    monkeypatch.delenv("EMBEDDING_PROVIDER", raising=False)
    monkeypatch.setenv("CODERAG_HIP_STORE_DTYPE", "f32")
    for dtype in ("f32", "bf16"):
        monkeypatch.setenv("CODERAG_HIP_STORE_DTYPE", dtype)
        if dtype == "bf16":
            continue      # (scores of the bf16 store are checked against the bf16 oracle in the search tests)
        asyncio.run(run_reference_database_scenario(QdrantManager(), CollectionName))


def test_golden_search_cases_on_hip_index(gpu):
    import numpy as np
    import coderag_amd  # noqa: F401
    from coderag_amd import ffi
    from tests.search_cases import CASES, make_case
    gold = np.load(__file__.rsplit("/", 1)[0] + "/golden/search_cases.npz")
    for name in CASES:
        c = make_case(name)
        for bf16 in (False, True):
            tag = f"{name}/{'bf16' if bf16 else 'f32'}"
            ncols = 0 if c.get("codes") is None else c["codes"].shape[1]
            idx = ffi.Index(768, ffi.DTYPE_BF16 if bf16 else ffi.DTYPE_F32, capacity_rows=max(64, len(c["x"])), n_code_cols=ncols)
            idx.append(c["x"], c.get("codes"))
            if c.get("alive") is not None:
                idx.tombstone(np.flatnonzero(c["alive"] == 0))
            s, r = idx.search(c["q"], c["k"], filters=c.get("filters"))
            assert np.array_equal(r, gold[f"{tag}/rows"]), tag
            assert np.array_equal(s.view(np.uint32), gold[f"{tag}/scores"].view(np.uint32)), tag
            idx.close()
