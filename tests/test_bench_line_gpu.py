"""The driver's contract for bench.py, on the GPU: ONE JSON line on stdout carrying the headline, `roofline`, `cpu_baseline`
and the sub-records -- run here on a small corpus with short CPU legs."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_the_contract_line(gpu):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--rows", "300000", "--steps", "3", "--warmup", "1", "--sub-steps", "3",
           "--check-rows", "20000", "--cpu-seconds", "0.3", "--embed-chunks", "600", "--e2e-texts", "300",
           "--legs", "filtered,wide,f32_store,embed,cpu"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=500, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    assert len(lines[0].encode()) + 1 <= 4096, len(lines[0])          # round 4's 22 KB line was cut by the driver's bounded tail and went unparsed

    def refuse(c):
        raise AssertionError(f"non-strict JSON constant {c} in the line")
    d = json.loads(lines[0], parse_constant=refuse)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0 and "workload" in d["config"]
    roof = d["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["achieved"] > 0 and 0 < roof["frac"] < 1.0
    cpu = d["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["sample"]
    assert d["parity"]["ids_bit_exact"] and d["parity"]["scores_bit_exact"], d["parity"]
    assert roof["whole_step_frac"] > 0 and roof["kernel"] and roof["kernel_ms"] > 0 and roof["algorithmic_bytes_per_launch"] > 0
    for leg in ("filtered", "wide", "f32_store", "embed"):          # one scalar + one fraction per leg in the line ...
        assert "error" not in d[leg], (leg, d[leg])
        assert d[leg]["value"] > 0 and d[leg]["frac"] > 0
        assert set(d[leg]) <= {"value", "unit", "ms", "frac", "bound", "step_frac", "ok"}, d[leg]
    for leg in ("filtered", "wide", "f32_store"):
        assert d[leg]["ok"] is True
    # ... and the full records in the sidecar the line names
    full = json.load(open(os.path.join(ROOT, d["detail"])))
    assert full["value"] == pytest.approx(d["value"], rel=1e-5) and full["steps"] == 3
    for leg in ("filtered", "wide", "f32_store", "embed"):
        assert full[leg]["roofline"]["frac"] == pytest.approx(d[leg]["frac"], rel=1e-3)
    assert full["embed"]["cpu_baseline"]["value"] > 0
    assert full["search_stats"]["batches"] >= 3
