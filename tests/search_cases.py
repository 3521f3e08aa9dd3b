"""Seeded search cases shared by the oracle tests (CPU), the GPU parity tests and tests/golden/gen_search_goldens.py."""
import numpy as np

D = 768
CASES = ("basic_1k", "ragged_33", "tiny_k_gt_n", "filters_4k", "zero_and_unit_vectors", "duplicates")


def make_case(name: str) -> dict:
    if name == "basic_1k":
        rng = np.random.default_rng(101)
        x = rng.standard_normal((1000, D), dtype=np.float32) * rng.uniform(0.1, 9.0, (1000, 1)).astype(np.float32)
        return dict(x=x, q=rng.standard_normal((5, D), dtype=np.float32), k=10)
    if name == "ragged_33":
        rng = np.random.default_rng(102)
        return dict(x=rng.standard_normal((33, D), dtype=np.float32), q=rng.standard_normal((1, D), dtype=np.float32), k=10)
    if name == "tiny_k_gt_n":
        rng = np.random.default_rng(103)
        return dict(x=rng.standard_normal((7, D), dtype=np.float32), q=rng.standard_normal((2, D), dtype=np.float32), k=10)
    if name == "filters_4k":
        rng = np.random.default_rng(104)
        n = 4000
        x = rng.standard_normal((n, D), dtype=np.float32)
        codes = np.stack([rng.integers(0, 3, n), rng.integers(0, 5, n)], axis=1).astype(np.int32)
        alive = (rng.random(n) > 0.2).astype(np.uint8)
        return dict(x=x, q=rng.standard_normal((4, D), dtype=np.float32), k=20, codes=codes, alive=alive, filters=[(0, 1), (1, 2)])
    if name == "zero_and_unit_vectors":
        rng = np.random.default_rng(105)
        x = rng.standard_normal((300, D), dtype=np.float32)
        x[3] = 0.0
        x[4] /= np.linalg.norm(x[4])
        x[5] = 1e-5 * x[5] / np.linalg.norm(x[5])
        q = rng.standard_normal((3, D), dtype=np.float32)
        q[2] = 0.0                       # a zero query scores 0 against everything: pure tie-break by row
        return dict(x=x, q=q, k=10, has_exact_ties=True)
    if name == "duplicates":
        rng = np.random.default_rng(106)
        base = rng.standard_normal((50, D), dtype=np.float32)
        x = np.concatenate([np.repeat(base[:1], 40, axis=0), base, np.repeat(base[1:2], 30, axis=0)])
        return dict(x=x, q=np.stack([base[0] + 0.01 * base[3], base[1]]), k=25, has_exact_ties=True)
    raise KeyError(name)
