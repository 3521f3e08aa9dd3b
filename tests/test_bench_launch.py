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
    assert sorted(rec["per_rank_local_rank"]) == list(range(8))
    assert len(lines[0]) < 4096
    full = json.load(open(os.path.join(ROOT, rec["detail"])))             # the sidecar holds the full per-rank device records
    assert [d["rank"] for d in full["per_rank_device"]] == list(range(8))


def test_world_size_mismatch_is_refused():
    p = _run("--gpus", "2", "--backend", "gloo", env_extra={"WORLD_SIZE": "3", "RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=3" in (p.stderr + p.stdout)


def test_parent_of_a_self_launch_never_imports_torch():
    """The launching parent must stay clear of HIP: it decides from argv + env alone and only spawns a child."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("def rehearse")]
    assert "import torch" not in head.replace("torch.distributed.run", "")


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_the_stdout_line_of_a_full_record_stays_under_4_kib_and_is_strict_json(tmp_path):
    """Round 4's driver record went unparsed: the one line had grown to 22 KB and the driver keeps a bounded tail.  The line is
    now distilled from the full record (which goes to a sidecar): here from round 4's committed full record, all legs present."""
    bench = _bench_module()
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_default_v6.json")))
    full["embed"]["roofline"]["frac"] = float("nan")                       # a NaN in a sub-record must not make the line non-strict
    r, w = os.pipe()
    bench.ROOT = str(tmp_path)                                            # sidecar into the test's directory
    bench.emit(full, w)
    os.close(w)
    text = os.read(r, 1 << 16).decode()
    os.close(r)
    assert text.endswith("\n") and text.count("\n") == 1 and len(text.encode()) <= 4096

    def refuse(c):
        raise AssertionError(c)
    line = json.loads(text, parse_constant=refuse)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline", "parity"):
        assert key in line, key
    roof = line["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "whole_step_frac", "bf16_scan"} <= set(roof)
    assert roof["bf16_scan"]["algorithmic_bytes_per_launch"] == 10_000_000 * 768 * 2 and roof["algorithmic_bytes_per_launch"] == 10_000_000 * 772
    assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"])
    for leg in bench.SHORT_LEGS:
        assert line[leg]["value"] > 0, leg
    assert line["embed"].get("frac") is None                               # the NaN became null / absent, not a bare NaN token
    side = json.load(open(tmp_path / line["detail"]))
    assert side["c2"]["parity"]["bf16_store"]["ids_bit_exact"] is True


def test_write_all_finishes_a_short_write(monkeypatch):
    bench = _bench_module()
    got = []

    def stingy(fd, data):
        got.append(bytes(data[:7]))
        return len(got[-1])
    monkeypatch.setattr(bench.os, "write", stingy)
    bench.write_all(1, b"x" * 100)
    assert b"".join(got) == b"x" * 100


def test_first_contact_checks_name_what_is_wrong_with_an_n_gpu_record():
    """Round-4 review, item 6: the first real 8-GPU record must fail loudly when it is not what it claims -- fewer ranks in the
    collectives than --gpus, two ranks on one GPU, a rank that nominated its batches differently from rank 0 or fell back."""
    bench = _bench_module()
    dev = [{"rank": r, "local_rank": r, "pci_bus_id": f"0000:{r:02x}:00.0"} for r in range(4)]
    nom = [[2, 0]] * 4
    assert bench.first_contact_errors(4, 4, dev, nom, True) == []
    assert any("3 distinct ranks" in e for e in bench.first_contact_errors(4, 3, dev, nom, True))
    twice = [dict(d) for d in dev]
    twice[3]["pci_bus_id"] = twice[1]["pci_bus_id"]
    assert any("ranks 1 and 3" in e and "same device" in e for e in bench.first_contact_errors(4, 4, twice, nom, True))
    assert bench.first_contact_errors(4, 4, twice, nom, True, shared_gpu=True) == []          # (the one-GPU rehearsal shares by design)
    assert any("rank 2 nominated" in e for e in bench.first_contact_errors(4, 4, dev, [[2, 0], [2, 0], [1, 0], [2, 0]], True))
    assert any("rank 1 fell back" in e for e in bench.first_contact_errors(4, 4, dev, [[2, 0], [2, 4], [2, 0], [2, 0]], True))
    assert bench.first_contact_errors(4, 4, dev, [[2, 1]] * 4, True) == []                   # (bit 0, buffers regrown once, is not a fallback)
    assert any("merged top-k" in e for e in bench.first_contact_errors(4, 4, dev, nom, False))
    no_bus = [{"rank": r, "local_rank": 0, "pci_bus_id": None} for r in range(2)]             # no bus ids: local ranks stand in
    assert any("same device" in e for e in bench.first_contact_errors(2, 2, no_bus, None, None))


def test_a_rank_that_dies_mid_run_takes_the_others_down_instead_of_hanging_them(tmp_path):
    """Each rank is a fresh child process launched the way a launcher would (RANK / WORLD_SIZE / MASTER_* in the environment);
    rank 1 is killed while the steps run; rank 0 must exit non-zero by itself, well inside the collective time-out."""
    import signal
    import socket
    import time
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, WORLD_SIZE="2", RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   CODERAG_BENCH_COLLECTIVE_TIMEOUT_S="30")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--rows", "4096",
                                       "--steps", "5000000", "--warmup", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    try:
        time.sleep(8.0)                                            # (torch import + rendezvous + some thousands of steps)
        assert procs[0].poll() is None and procs[1].poll() is None, "the rehearsal ended before the kill"
        procs[1].send_signal(signal.SIGKILL)
        t0 = time.time()
        # (were rank 1 still importing torch when it was killed -- a cold container -- rank 0 gives up in the rendezvous instead, after the
        # 30 s collective time-out set above: non-zero all the same, hence the generous limits)
        out, err = procs[0].communicate(timeout=240)
        assert procs[0].returncode not in (0, None), (procs[0].returncode, err[-500:])
        assert time.time() - t0 < 180
        assert out.strip() == ""                                   # no line that could be taken for a measurement
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
            p.wait()
