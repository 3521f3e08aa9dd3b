"""The CPU oracle against the committed fixtures and an independent fp64 evaluation (no GPU)."""
import numpy as np
import pytest

from oracle import search as orc
from tests.search_cases import CASES, make_case

GOLD = np.load(__file__.rsplit("/", 1)[0] + "/golden/search_cases.npz")


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_golden(name, bf16):
    c = make_case(name)
    tag = f"{name}/{'bf16' if bf16 else 'f32'}"
    assert np.isclose(c["x"].astype(np.float64).sum(), GOLD[f"{tag}/xsum"], rtol=0, atol=1e-6), "case inputs drifted"
    s, r = orc.cosine_search(c["x"], c["q"], c["k"], bf16=bf16, alive=c.get("alive"), codes=c.get("codes"), filters=c.get("filters"))
    assert np.array_equal(r, GOLD[f"{tag}/rows"])
    assert np.array_equal(s.view(np.uint32), GOLD[f"{tag}/scores"].view(np.uint32))


def test_preprocess_follows_qdrant_rules():
    rng = np.random.default_rng(0)
    v = rng.standard_normal((4, 768)).astype(np.float32) * 7
    v[1] = 0
    v[2] = v[2] / np.linalg.norm(v[2].astype(np.float64))
    v[3] = 1e-5 * v[3] / np.linalg.norm(v[3])
    out = orc.preprocess(v)
    assert abs(np.linalg.norm(out[0].astype(np.float64)) - 1) < 1e-6
    assert np.array_equal(out[1], v[1])                      # zero stays zero
    assert np.array_equal(out[3], v[3])                      # squared length < f32 epsilon: untouched
    len2 = np.float32(0)
    for x in v[2]:
        len2 = np.float32(len2 + np.float32(x * x))
    if abs(float(len2) - 1.0) <= 1e-6:                       # already normalised: untouched
        assert np.array_equal(out[2], v[2])
    bf = orc.preprocess(v, to_bf16=True)
    assert np.all((bf.view(np.uint32) & 0xFFFF) == 0)
    assert np.abs(bf - out).max() <= np.abs(out).max() * 2 ** -8


def test_oracle_vs_fp64_larger_random():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((20000, 768)).astype(np.float32)
    q = rng.standard_normal((8, 768)).astype(np.float32)
    xp, qp = orc.preprocess(x), orc.preprocess(q)
    s, r = orc.search(xp, qp, 50)
    s64, r64 = orc.search_fp64(xp, qp, 50)
    assert np.abs(s - s64).max() < 2e-6
    assert np.mean(r == r64) > 0.99                          # only sub-1e-7 near-ties may swap
    assert [set(a) for a in r] == [set(b) for b in r64] or np.mean([len(set(a) & set(b)) for a, b in zip(r, r64)]) >= 49.5
    sb, rb = orc.search_blas(xp, qp, 50)
    assert np.abs(sb - s).max() < 2e-6


def test_merge_topk_equals_global_search():
    rng = np.random.default_rng(4)
    x = rng.standard_normal((3000, 768)).astype(np.float32)
    q = rng.standard_normal((6, 768)).astype(np.float32)
    xp, qp = orc.preprocess(x), orc.preprocess(q)
    parts = [orc.search(xp[a:b], qp, 30) for a, b in ((0, 1000), (1000, 1010), (1010, 3000))]
    scores = np.stack([p[0] for p in parts])
    rows = np.stack([np.where(p[1] >= 0, p[1] + off, -1) for p, off in zip(parts, (0, 1000, 1010))])
    ms, mr = orc.merge_topk(scores, rows)
    gs, gr = orc.search(xp, qp, 30)
    assert np.array_equal(mr, gr) and np.array_equal(ms, gs)
