"""`python bench.py --gpus N` must start its own N ranks (the driver's multi-GPU command shape) -- exercised here without a
GPU through the launcher rehearsal: same self-launch (torch.distributed.run as a child, parent never touches HIP), same
rendezvous, the exchange-record all-gather on gloo, max-over-ranks timing, ONE JSON line from rank 0."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=300, env=env)


@pytest.mark.parametrize("n,scaling", [(2, "weak"), (4, "strong")])
def test_self_launch_prints_one_json_line_with_all_ranks(n, scaling):
    p = _run("--gpus", str(n), "--backend", "gloo", "--rows", "4096", "--steps", "3", "--warmup", "1", "--scaling", scaling)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout                      # rank 0 only, nothing else on stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == n and rec["collective_ranks"] == n and rec["exchange_layout_ok"] is True
    assert rec["value"] is None and rec["backend"] == "gloo"            # a rehearsal never looks like a measurement
    assert len(rec["per_rank_step_ms"]) == n and rec["scaling"] == scaling
    assert rec["rows_per_gpu"] == (4096 // n if scaling == "strong" else 4096)
    assert rec["legs_rehearsed"].get("c2") is True                      # default legs at N > 1 now include c2 (per-rank embed -> local shard -> sharded top-k)
    assert len(rec["per_rank_nomination"]) == n and len(rec["per_rank_fallback_used"]) == n


def test_eight_ranks_strong_scaling_with_the_config5_and_embed_collectives():
    """The shape of the first real 8-GPU record (SCALE_rNN): 8 ranks, --scaling strong, the config5 and embed legs beside the
    headline -- every collective those legs issue runs (gloo) and its layout is checked; per-rank device records are there."""
    p = _run("--gpus", "8", "--backend", "gloo", "--rows", "80000", "--steps", "2", "--warmup", "1", "--scaling", "strong", "--legs", "config5,embed")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["collective_ranks"] == 8 and rec["exchange_layout_ok"] is True
    assert rec["legs_rehearsed"] == {"config5": True, "embed": True}
    # the first real SCALE record must show at a glance whether every rank took the int8 pass: [mode, fallback_used] of every
    # rank travel in one all-gather (here the rehearsal's stand-ins, -1 - rank and rank, in rank order)
    assert rec["per_rank_nomination"] == [-1 - r for r in range(8)] and rec["per_rank_fallback_used"] == list(range(8))
    assert rec["rows_per_gpu"] == 10000 and rec["scaling"] == "strong" and len(rec["per_rank_step_ms"]) == 8
    assert [d["rank"] for d in rec["per_rank_device"]] == list(range(8)) and sorted(d["local_rank"] for d in rec["per_rank_device"]) == list(range(8))


def test_world_size_mismatch_is_refused():
    p = _run("--gpus", "2", "--backend", "gloo", env_extra={"WORLD_SIZE": "3", "RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=3" in (p.stderr + p.stdout)


def test_parent_of_a_self_launch_never_imports_torch():
    """The launching parent must stay clear of HIP: it decides from argv + env alone and only spawns a child."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("def rehearse")]
    assert "import torch" not in head.replace("torch.distributed.run", "")
