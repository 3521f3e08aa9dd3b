#!/usr/bin/env python3
"""Freeze small search fixtures: tests/golden/search_cases.npz.

Inputs are regenerated from seeds by tests/search_cases.py (shared by this script, the CPU oracle tests and the
GPU parity tests); the expected outputs stored here come from an independent numpy fp64 evaluation of the
reference semantics (normalise on insert and on query, dot product, descending, ties by lower row) and are only
kept where the fp64 ranking has no near-tie closer than 1e-6 at any returned rank, so the f32 oracle and kernel
must reproduce the ids exactly.  Scores are stored as the C oracle's f32 values (pinned against fp64 to 2e-6).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import search as orc  # noqa: E402
from tests.search_cases import CASES, make_case  # noqa: E402

out = {}
for name in CASES:
    c = make_case(name)
    for bf16 in (False, True):
        xp, qp = orc.preprocess(c["x"], bf16), orc.preprocess(c["q"], bf16)
        alive = c.get("alive")
        mask = np.ones(len(xp), bool) if alive is None else alive.astype(bool)
        for col, val in c.get("filters", []):
            mask &= c["codes"][:, col] == val
        s64, r64 = orc.search_fp64(xp, qp, c["k"], alive=mask)
        s32, r32 = orc.search(xp, qp, c["k"], alive=c.get("alive"), codes=c.get("codes"), filters=c.get("filters"))
        gaps = np.abs(np.diff(np.where(np.isfinite(s64), s64, 0.0), axis=1))
        tag = f"{name}/{'bf16' if bf16 else 'f32'}"
        exact_dupes = c.get("has_exact_ties", False)
        if not exact_dupes:
            assert np.array_equal(r64, r32), f"{tag}: fp64 and f32 oracle disagree on ids"
        assert np.nanmax(np.abs(np.where(r32 >= 0, s32 - s64, 0))) < 2e-6, tag
        out[f"{tag}/rows"] = r32
        out[f"{tag}/scores"] = s32
        out[f"{tag}/xsum"] = np.float64(c["x"].astype(np.float64).sum())
        print(tag, "ok", r32.shape, "min gap", float(gaps[np.isfinite(gaps)].min()) if gaps.size else None)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "search_cases.npz"), **out)
print("wrote tests/golden/search_cases.npz")
